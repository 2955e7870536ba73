#pragma once
#include <microscopes_amd/hip_models.hpp>
