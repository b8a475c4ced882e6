#!/bin/bash
# One parameterised runner for the GPU box (replaces the per-experiment gpu_*.sh scripts).
#   tools/gpu.sh test [pytest-args...]          all -m gpu tests (or a selection), log under gpurun_out/
#   tools/gpu.sh ab <variant> <variant> ...     interleaved A/B of library builds inside ONE box
#                                               ("default" = libvoicesynth.so, else libvoicesynth_<v>.so from
#                                               `make variant NAME=<v> DEFS=...`); AB_CONFIGS="3 5 4" AB_REPS=3
#   tools/gpu.sh bench [bench.py args...]       bench.py, JSON line to gpurun_out/bench.json
#   tools/gpu.sh prof <tag> [bench.py args...]  rocprofv3 kernel trace of bench.py -> gpurun_out/prof_<tag>/
#   tools/gpu.sh pmc <tag> <counters...>        one rocprofv3 --pmc pass of bench.py per invocation
# Steps may be chained:  tools/gpu.sh test -- ab default r2 -- bench
# Boxes differ by up to 10 % in sustained clock, so variants are only ever compared within one call.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
cd "$(dirname "$0")/.."

run_step() {
  local step="$1"; shift
  case "$step" in
    test)
      timeout -k 10 900 python -m pytest tests -m gpu -x -q "$@" > gpurun_out/pytest_gpu.log 2>&1
      local rc=$?; tail -12 gpurun_out/pytest_gpu.log; return $rc ;;
    ab)
      : > gpurun_out/ab.log
      for cfg in ${AB_CONFIGS:-3}; do
        local lanes=65536; [ "$cfg" = 4 ] && lanes=32768; [ "$cfg" = 2 ] && lanes=1024
        for rep in $(seq 1 ${AB_REPS:-3}); do for v in "$@"; do
          local lib=libvoicesynth_$v.so; [ "$v" = default ] && lib=libvoicesynth.so
          echo "== config $cfg rep $rep $v" | tee -a gpurun_out/ab.log
          VS_LIB=$lib timeout -k 10 180 python tools/quick_bench.py $cfg $lanes 5 2>&1 | grep -E "exact/synth|fma/synth|Error|error" | tee -a gpurun_out/ab.log || return 1
        done; done
      done ;;
    bench)
      timeout -k 10 600 python bench.py "$@" > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; return 1; }
      cat gpurun_out/bench.json ;;
    prof)
      local tag="$1"; shift
      ( cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_$tag" -o "$tag" --output-format csv -- python3 "$OLDPWD/bench.py" "$@" > "$OLDPWD/gpurun_out/prof_$tag.json" 2> "$OLDPWD/gpurun_out/prof_$tag.err" ) || { tail -5 gpurun_out/prof_$tag.err; return 1; }
      find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1 | xargs -r head -8 ;;
    pmc)
      local tag="$1"; shift
      ( cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --pmc "$@" -d "$OLDPWD/gpurun_out/pmc_$tag" -o "$tag" --output-format csv -- python3 "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$OLDPWD/gpurun_out/pmc_$tag.json" 2> "$OLDPWD/gpurun_out/pmc_$tag.err" ) || { tail -5 gpurun_out/pmc_$tag.err; return 1; } ;;
    *) echo "unknown step $step"; return 2 ;;
  esac
}

args=()
for a in "$@" --; do
  if [ "$a" = "--" ]; then
    [ ${#args[@]} -gt 0 ] && { run_step "${args[@]}" || exit $?; }
    args=()
  else
    args+=("$a")
  fi
done
