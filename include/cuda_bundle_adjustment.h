// Drop-in header name of the reference (include/cuda_bundle_adjustment.h): forwards to the MI355X build.
#pragma once
#include "cuba/bundle_adjustment.hpp"
