/** @file problem.hxx  problem_t lives with enactor_t in framework/bsp.hxx. */
#pragma once
#include <gunrock/framework/bsp.hxx>
