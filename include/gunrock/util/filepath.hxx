/** @file filepath.hxx  Reference include path (util/filepath.hxx:18-27): extract_filename / extract_dataset / is_market / is_binary_csr live in io/matrix_market.hxx. */
#pragma once
#include <gunrock/io/matrix_market.hxx>
