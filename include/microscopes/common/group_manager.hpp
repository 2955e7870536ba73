#pragma once
#include <microscopes_amd/group_manager.hpp>
