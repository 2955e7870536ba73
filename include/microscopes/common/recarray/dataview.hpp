#pragma once
#include <microscopes_amd/recarray.hpp>
