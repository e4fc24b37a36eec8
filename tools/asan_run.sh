#!/bin/bash
# Run a test selection against the sanitizer build of the host side (make -C ntjoin_amd/csrc asan | tsan):
#   tools/asan_run.sh [asan|tsan] <pytest arguments>      e.g.  tools/asan_run.sh asan tests -m "not gpu" -x -q
# python itself is not instrumented, so the sanitizer runtime is preloaded; leak checking is off (the interpreter never frees
# everything), the library's own reports still abort the run.
kind=${1:-asan}; shift
root=$(cd "$(dirname "$0")/.." && pwd)
rt=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.${kind}-x86_64.so)
[ -f "$root/ntjoin_amd/lib_$kind/libntjoin_mx.so" ] || make -C "$root/ntjoin_amd/csrc" $kind || exit 1
export MXG_NO_DETACH=1  # (mxgraph in one process: a sanitizer report comes with the real exit of the process that did the work)
export MXG_LIB_DIR=$root/ntjoin_amd/lib_$kind MXG_BIN_DIR=$root/ntjoin_amd/bin_$kind
# (tsan: the uninstrumented HIP / HSA runtimes are suppressed, tools/tsan.supp; a race in this library's own code still ends the
# process that shows it with status 66, which fails the test that started it)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1:verify_asan_link_order=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
# (asan: the pinned pool in one hipHostMalloc.  With the pool registered piece by piece -- the default -- ROCm's AddressSanitizer runtime
# aborts inside itself when the process ENDS: libamdhip64's finalizer frees an object, the quarantine recycles an older chunk of the
# runtime's device allocator, and that allocator CHECKs "device runtime not unloaded" (sanitizer_allocator_device.h:125): no report
# about this library's memory, and nothing of it without the sanitizer.  The thread that registers the pieces is covered by the tsan run.)
[ "$kind" = asan ] && export MXG_PIN_MALLOC=1
export TSAN_OPTIONS=${TSAN_OPTIONS:-"suppressions=$root/tools/tsan.supp:halt_on_error=0:exitcode=66:report_signal_unsafe=0:second_deadlock_stack=1"}
# (tests that use torch for device buffers: torch.cuda's lazy init dlopens libcaffe2_nvrtc.so by its bare name, and under the
# preloaded sanitizer runtime -- whose dlopen interceptor is the caller then -- torch's own RUNPATH no longer applies)
export LD_LIBRARY_PATH=$(python -c "import importlib.util, os; print(os.path.join(os.path.dirname(importlib.util.find_spec('torch').origin), 'lib'))"):$LD_LIBRARY_PATH
cd "$root" && LD_PRELOAD=$rt python -m pytest "$@"
