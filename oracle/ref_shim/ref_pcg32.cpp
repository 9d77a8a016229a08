// TEST INFRASTRUCTURE ONLY. Host build of the reference's pcg32 (ops/op_include/pcg32/pcg32.h) for known-answer tests.
#include "pcg32.h"
#define EXP extern "C" __attribute__((visibility("default")))
EXP void ref_pcg32_seed(uint64_t initstate, uint64_t initseq, uint64_t *st) { pcg32 r(initstate, initseq); st[0] = r.state; st[1] = r.inc; }
EXP uint32_t ref_pcg32_next_uint(uint64_t *st) { pcg32 r; r.state = st[0]; r.inc = st[1]; uint32_t v = r.next_uint(); st[0] = r.state; return v; }
EXP float ref_pcg32_next_float(uint64_t *st) { pcg32 r; r.state = st[0]; r.inc = st[1]; float v = r.next_float(); st[0] = r.state; return v; }
EXP void ref_pcg32_advance(uint64_t *st, int64_t delta) { pcg32 r; r.state = st[0]; r.inc = st[1]; r.advance(delta); st[0] = r.state; }
