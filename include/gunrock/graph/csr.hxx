/** @file csr.hxx  Reference include path (graph/csr.hxx:31-234): graph_csr_t lives in graph/graph.hxx. */
#pragma once
#include <gunrock/graph/graph.hxx>
