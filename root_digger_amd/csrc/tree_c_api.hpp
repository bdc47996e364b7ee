// access to the C++ object behind a rdamd_tree_t handle (for model_c_api.cpp)
#pragma once
#include "tree.hpp"
const rdamd::rooted_tree_t &rdamd_tree_cpp(const rdamd_tree_t *t);
