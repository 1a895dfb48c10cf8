/** @file filter.hxx  Reference include path (operators/filter/filter.hxx:59-152). */
#pragma once
#include <gunrock/framework/operators/filter.hxx>
