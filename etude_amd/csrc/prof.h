// Per-kernel HIP-event profiler of the library's own launches (off by default; bench.py turns it on to
// measure the dominant kernel's average launch duration live, on the stream the kernel runs on).
#pragma once
#include "common.h"

bool prof_enabled();
hipEvent_t prof_begin(hipStream_t st);                       // null when the profiler is off
void prof_end(const char* name, hipEvent_t start, hipStream_t st, double flops, double bytes);

// Diagnostic: ETD_LAUNCH_LOG=1 prints every launcher's name to stderr before its launch and synchronises the stream behind it (the last name printed before a
// GPU fault is the faulting launch; off: one relaxed load per launch).  Not compatible with stream capture: use with ETD_NO_GRAPH=1.
bool launch_log_on();
void launch_log(const char* name, hipStream_t st, bool after);

struct ProfScope {   // the start event lives in the scope object, so concurrent launches from several host threads do not mix
  const char* name; hipStream_t st; double flops, bytes; hipEvent_t start;
  ProfScope(const char* n, hipStream_t s, double f = 0, double b = 0) : name(n), st(s), flops(f), bytes(b), start(prof_begin(s)) { if (launch_log_on()) launch_log(n, s, false); }
  ~ProfScope() { if (start) prof_end(name, start, st, flops, bytes); if (launch_log_on()) launch_log(name, st, true); }
};

// Diagnostic (tools/probe_race.py): ETD_EXT_ONLY=name[,name...] makes every OTHER Extract-stage launcher return without launching
// (outputs are then garbage; buffer shapes do not change, so what does run stays in bounds).  Unset: nothing is skipped.
bool launch_skipped(const char* name);
#define ETD_LAUNCH_FILTER(name) do { if (launch_skipped(name)) return ETD_OK; } while (0)
