/** @file framework.hxx  Frontier + problem + enactor + operators in one include. */
#pragma once
#include <gunrock/framework/frontier.hxx>
#include <gunrock/framework/bitmap_frontier.hxx>
#include <gunrock/framework/problem.hxx>
#include <gunrock/framework/enactor.hxx>
#include <gunrock/framework/operators/operators.hxx>
