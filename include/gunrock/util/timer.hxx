/** @file timer.hxx  Reference include path (util/timer.hxx:16-49): util::timer_t lives in hip/context.hxx (it records on the context's stream). */
#pragma once
#include <gunrock/hip/context.hxx>
