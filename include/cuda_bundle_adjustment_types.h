// Drop-in header name of the reference (include/cuda_bundle_adjustment_types.h): forwards to the MI355X build.
#pragma once
#include "cuba/ba_types.hpp"
