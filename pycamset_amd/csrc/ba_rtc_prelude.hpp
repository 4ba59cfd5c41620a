// ba_rtc_prelude.hpp — what <cstdint> would have given, when a chain is compiled by hiprtc (no host headers there)
#pragma once
typedef signed char int8_t;
typedef unsigned char uint8_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long long int64_t;
typedef unsigned long long uint64_t;
typedef unsigned long uintptr_t;
