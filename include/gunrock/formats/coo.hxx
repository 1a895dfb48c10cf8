/** @file coo.hxx  Reference include path (formats/coo.hxx:21-46): format::coo_t lives in formats/formats.hxx. */
#pragma once
#include <gunrock/formats/formats.hxx>
