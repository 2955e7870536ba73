#pragma once
#include <microscopes_amd/types.hpp>
