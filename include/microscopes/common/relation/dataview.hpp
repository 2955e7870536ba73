#pragma once
#include <microscopes_amd/relation.hpp>
