/** @file enactor.hxx  enactor_t / enactor_properties_t live in framework/bsp.hxx. */
#pragma once
#include <gunrock/framework/bsp.hxx>
