#pragma once
// thread::load / thread::store live with the atomics facade.
#include <gunrock/util/math.hxx>
