/** @file properties.hxx  Reference include path (graph/properties.hxx): graph_properties_t lives in graph/graph.hxx. */
#pragma once
#include <gunrock/graph/graph.hxx>
