#!/bin/bash
# Run a test selection against the sanitizer build of the host side (make -C ntjoin_amd/csrc asan | tsan):
#   tools/asan_run.sh [asan|tsan] <pytest arguments>      e.g.  tools/asan_run.sh asan tests -m "not gpu" -x -q
# python itself is not instrumented, so the sanitizer runtime is preloaded; leak checking is off (the interpreter never frees
# everything), the library's own reports still abort the run.
kind=${1:-asan}; shift
root=$(cd "$(dirname "$0")/.." && pwd)
rt=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.${kind}-x86_64.so)
[ -f "$root/ntjoin_amd/lib_$kind/libntjoin_mx.so" ] || make -C "$root/ntjoin_amd/csrc" $kind || exit 1
export MXG_LIB_DIR=$root/ntjoin_amd/lib_$kind MXG_BIN_DIR=$root/ntjoin_amd/bin_$kind
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1:verify_asan_link_order=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 TSAN_OPTIONS=halt_on_error=1
cd "$root" && LD_PRELOAD=$rt python -m pytest "$@"
