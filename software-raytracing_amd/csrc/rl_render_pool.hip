// The pool schedule of the megakernel (k_trace_pool) as a translation unit of its own: the same source as rl_render.hip, which then only declares
// that kernel, compiled with LLVM's "iterative-ilp" scheduler strategy (Makefile).  Measured on MI355X, same bits: 298 k-triangle scene 48.9 -> 44.8 ms,
// colonnade 439 -> 413 ms, 2.36 M triangles at 4K 149.7 -> 130.6 ms; k_trace (Cornell) prefers the default strategy by 0.4 %, hence two units.
#define RL_TU_POOL 1
#include "rl_render.hip"
