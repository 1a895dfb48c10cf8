/** @file uniquify.hxx  Reference include path (operators/uniquify/uniquify.hxx:15-42): uniquify::execute lives in operators/filter.hxx. */
#pragma once
#include <gunrock/framework/operators/filter.hxx>
