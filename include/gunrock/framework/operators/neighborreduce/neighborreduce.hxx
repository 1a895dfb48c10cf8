/** @file neighborreduce.hxx  Reference include path (operators/neighborreduce/neighborreduce.hxx:55-101). */
#pragma once
#include <gunrock/framework/operators/neighborreduce.hxx>
