#pragma once
#include <microscopes_amd/timer.hpp>
