/** @file context.hxx  Reference include path (cuda/context.hxx:54-206): gcuda::standard_context_t / multi_context_t live in hip/context.hxx. */
#pragma once
#include <gunrock/hip/context.hxx>
