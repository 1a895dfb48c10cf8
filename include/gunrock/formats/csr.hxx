/** @file csr.hxx  Reference include path (formats/csr.hxx:25-238): format::csr_t lives in formats/formats.hxx. */
#pragma once
#include <gunrock/formats/formats.hxx>
