/** @file error.hxx  Reference include path (error.hxx:21-46): error::exception_t / throw_if_exception live in hip/runtime.hxx. */
#pragma once
#include <gunrock/hip/runtime.hxx>
