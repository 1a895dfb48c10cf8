/** @file atomic_functions.hxx  Reference include path (cuda/atomic_functions.hxx:36-123): the float/double atomic min/max are in util/math.hxx (one integer atomic on the bit pattern, no CAS loop). */
#pragma once
#include <gunrock/util/math.hxx>
