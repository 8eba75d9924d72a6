// Host-only part of the library's common header: error reporting and argument checks.  No HIP include, so that csrc/host.cpp (the
// TFRecord / protobuf walk over untrusted file bytes) also builds as plain C++ under -fsanitize=address,undefined
// (tests/test_host_sanitized.py).
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/las_hip.h"

void las_set_error(const char* fmt, ...);

#define LAS_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      las_set_error(__VA_ARGS__);         \
      return LAS_ERR_ARG;                 \
    }                                     \
  } while (0)
