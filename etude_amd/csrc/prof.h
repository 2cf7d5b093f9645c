// Per-kernel HIP-event profiler of the library's own launches (off by default; bench.py turns it on to
// measure the dominant kernel's average launch duration live, on the stream the kernel runs on).
#pragma once
#include "common.h"

bool prof_enabled();
void prof_begin(const char* name, hipStream_t st);
void prof_end(const char* name, hipStream_t st, double flops, double bytes);

struct ProfScope {
  const char* name; hipStream_t st; double flops, bytes;
  ProfScope(const char* n, hipStream_t s, double f = 0, double b = 0) : name(n), st(s), flops(f), bytes(b) { prof_begin(name, st); }
  ~ProfScope() { prof_end(name, st, flops, bytes); }
};
