/** @file framework.hxx  Frontier + problem + enactor + operators in one include. */
#pragma once
#include <gunrock/framework/frontier.hxx>
#include <gunrock/framework/bitmap_frontier.hxx>
#include <gunrock/framework/bsp.hxx>
#include <gunrock/framework/operators/operators.hxx>
