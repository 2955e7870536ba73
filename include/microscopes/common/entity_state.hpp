#pragma once
#include <microscopes_amd/entity_state.hpp>
