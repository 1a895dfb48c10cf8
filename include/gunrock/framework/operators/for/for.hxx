/** @file for.hxx  Reference include path (operators/for/for.hxx:28-96): parallel_for::execute lives in operators/filter.hxx. */
#pragma once
#include <gunrock/framework/operators/filter.hxx>
