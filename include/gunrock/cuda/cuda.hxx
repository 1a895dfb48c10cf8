/** @file cuda.hxx  Reference include path (cuda/cuda.hxx:19-27): the whole device layer. */
#pragma once
#include <gunrock/hip/runtime.hxx>
#include <gunrock/hip/context.hxx>
#include <gunrock/hip/launch_box.hxx>
