/** @file batch.hxx  Reference include path (operators/batch/batch.hxx:61-79). */
#pragma once
#include <gunrock/framework/operators/batch.hxx>
