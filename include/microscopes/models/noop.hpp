#pragma once
#include <microscopes_amd/plugin.hpp>
