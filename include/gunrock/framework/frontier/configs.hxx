/** @file configs.hxx  Reference include path (framework/frontier/configs.hxx:20-24): frontier_kind_t / frontier_view_t. */
#pragma once
#include <gunrock/framework/frontier.hxx>
