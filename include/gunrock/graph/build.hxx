/** @file build.hxx  Reference include path (graph/build.hxx:26-52): graph::build::from_csr lives in graph/graph.hxx. */
#pragma once
#include <gunrock/graph/graph.hxx>
