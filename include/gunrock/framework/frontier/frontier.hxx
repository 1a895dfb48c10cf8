/** @file frontier.hxx  Reference include path (framework/frontier/frontier.hxx:33-148). */
#pragma once
#include <gunrock/framework/frontier.hxx>
