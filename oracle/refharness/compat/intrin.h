// MSVC header name -> GCC equivalent (reference Core/Math.h:5 includes <intrin.h> for _mm_cvt_ss2si).
#pragma once
#include <x86intrin.h>
