/** @file operators.hxx  Every operator of the engine. */
#pragma once
#include <gunrock/framework/operators/configs.hxx>
#include <gunrock/framework/operators/advance.hxx>
#include <gunrock/framework/operators/filter.hxx>
#include <gunrock/framework/operators/batch.hxx>
#include <gunrock/framework/operators/neighborreduce.hxx>
