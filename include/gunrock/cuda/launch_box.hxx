/** @file launch_box.hxx  Reference include path (cuda/launch_box.hxx:116-335): gcuda::launch_box lives in hip/launch_box.hxx. */
#pragma once
#include <gunrock/hip/launch_box.hxx>
