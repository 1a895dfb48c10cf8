/** @file advance.hxx  Reference include path (operators/advance/advance.hxx:91-221). */
#pragma once
#include <gunrock/framework/operators/advance.hxx>
