/** @file vector_frontier.hxx  Reference include path (framework/frontier/vector_frontier.hxx:28-256). */
#pragma once
#include <gunrock/framework/frontier.hxx>
