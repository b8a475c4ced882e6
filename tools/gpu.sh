#!/bin/bash
# One parameterised runner for the GPU box (replaces the per-experiment gpu_*.sh scripts).
#   tools/gpu.sh test [pytest-args...]          all -m gpu tests (or a selection), log under gpurun_out/
#   tools/gpu.sh ab <variant> <variant> ...     interleaved A/B of library builds inside ONE box
#                                               ("default" = libvoicesynth.so, else libvoicesynth_<v>.so from
#                                               `make variant NAME=<v> DEFS=...`); AB_CONFIGS="3 5 4" AB_REPS=3
#   tools/gpu.sh bench [bench.py args...]       bench.py, JSON line to gpurun_out/bench.json
#   tools/gpu.sh prof <tag> [bench.py args...]  rocprofv3 kernel trace of bench.py -> gpurun_out/prof_<tag>/
#   tools/gpu.sh pmc <tag> <counters...>        one rocprofv3 --pmc pass of bench.py per invocation
#   tools/gpu.sh traffic <key> <kernel> [args]  FETCH_SIZE + WRITE_SIZE passes -> profiles/pmc_traffic.json[key]
#   tools/gpu.sh sq <key> <kernel> [args]       SQ counter passes -> profiles/pmc_valu.json[key]
#   tools/gpu.sh smoke                          __graft_entry__.smoke()
#   tools/gpu.sh roles [config] [lanes]         two-role against three-role kernel (+ timing-only builds)
#   tools/gpu.sh sweep <roles> [config] [lanes] generator round-start knobs
#   tools/gpu.sh diag                           phase breakdown from the diagnostic build
#   tools/gpu.sh full <tag>                     the round's evidence pass (copy the summaries into profiles/)
#   tools/gpu.sh configs <tag>                  bench.py --config 2 / 4 / 5 under rocprofv3 --kernel-trace --stats
#   tools/gpu.sh pmcset 3|3n|2|4|5|f            traffic + SQ counter passes of one configuration's keys (exact, fma[, f32]); f = the f32 / vowel -n rows left over
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
      ( cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --pmc "$@" -d "$OLDPWD/gpurun_out/pmc_$tag" -o "$tag" --output-format csv -- python3 "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > "$OLDPWD/gpurun_out/pmc_$tag.json" 2> "$OLDPWD/gpurun_out/pmc_$tag.err" ) || { tail -5 gpurun_out/pmc_$tag.err; return 1; } ;;
    smoke)
      timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
      local rc=$?; tail -3 gpurun_out/smoke.log; return $rc ;;
    traffic)
      # HBM bytes per launch: two SEPARATE --pmc passes (FETCH_SIZE, WRITE_SIZE), then profiles/pmc_traffic.json
      #   tools/gpu.sh traffic <key> <kernel substring> [bench.py args...]      VS_LIB selects a variant library
      local key="$1" kern="$2"; shift 2
      for C in FETCH_SIZE WRITE_SIZE; do
        ( cd /tmp && timeout -k 10 600 rocprofv3 --pmc $C --output-format csv -d "$OLDPWD/gpurun_out/pmc_$C" -o bench -- python3 "$OLDPWD/bench.py" --no-cpu-baseline --no-other-configs --steps 5 --warmup 2 "$@" > "$OLDPWD/gpurun_out/pmc_$C.log" 2>&1 ) || { tail -5 gpurun_out/pmc_$C.log; return 1; }
      done
      python tools/summarize_pmc.py traffic gpurun_out/pmc_FETCH_SIZE/bench_counter_collection.csv gpurun_out/pmc_WRITE_SIZE/bench_counter_collection.csv "$key" "$kern" || return 1
      cp profiles/pmc_traffic.json gpurun_out/ ;;   # only gpurun_out/ travels back: copy it into profiles/ by hand
    sq)
      # SQ counters per launch (two passes), then profiles/pmc_valu.json:  tools/gpu.sh sq <key> <kernel substring> [bench.py args...]
      local key="$1" kern="$2"; shift 2
      ( cd /tmp && timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OLDPWD/gpurun_out/pmc_sq1" -o bench -- python3 "$OLDPWD/bench.py" --no-cpu-baseline --no-other-configs --steps 5 --warmup 2 "$@" > "$OLDPWD/gpurun_out/pmc_sq1.log" 2>&1 ) || { tail -5 gpurun_out/pmc_sq1.log; return 1; }
      ( cd /tmp && timeout -k 10 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d "$OLDPWD/gpurun_out/pmc_sq2" -o bench -- python3 "$OLDPWD/bench.py" --no-cpu-baseline --no-other-configs --steps 5 --warmup 2 "$@" > "$OLDPWD/gpurun_out/pmc_sq2.log" 2>&1 ) || { tail -5 gpurun_out/pmc_sq2.log; return 1; }
      python tools/summarize_pmc.py valu gpurun_out/pmc_sq1/bench_counter_collection.csv gpurun_out/pmc_sq2/bench_counter_collection.csv "$key" --kernel "$kern" || return 1
      cp profiles/pmc_valu.json gpurun_out/ ;;
    roles)
      # two-role against three-role wave-specialised kernel (and the timing-only builds, if present:
      # `make variant NAME=gonly DEFS=-DVS_TIMING_GENERATOR_ONLY`, `NAME=fonly DEFS=-DVS_TIMING_FILTER_ONLY`)
      local cfg=${1:-3} lanes=${2:-65536}
      for rep in 1 2; do for lib in libvoicesynth.so libvoicesynth_gonly.so libvoicesynth_fonly.so; do
        [ -f voice_synth_amd/lib/$lib ] || continue
        for roles in 2 3; do
          echo "== rep $rep $lib roles $roles"
          VS_LIB=$lib VS_DEBUG_TUNING=1 VS_WS_ROLES=$roles timeout -k 10 120 python tools/quick_bench.py $cfg $lanes 5 | grep -E "exact/synth|fma/synth"
        done
      done; done ;;
    sweep)
      # round-start knobs of the generator (vs_tuning.gen_min x gen_low):  tools/gpu.sh sweep <roles> [config] [lanes]
      local roles=${1:-3} cfg=${2:-3} lanes=${3:-65536}
      for gm in ${GEN_MINS:-32 48 56 64}; do for gl in ${GEN_LOWS:-32 48 72}; do
        echo -n "roles $roles gen_min $gm gen_low $gl: "
        VS_DEBUG_TUNING=1 VS_WS_ROLES=$roles VS_GEN_MIN=$gm VS_GEN_LOW=$gl timeout -k 10 120 python tools/quick_bench.py $cfg $lanes 4 | grep -E "exact/synth|fma/synth" | awk '{printf "%s %s ms   ", $1, $2}'; echo
      done; done ;;
    diag)
      # phase breakdown of the wave-specialised kernels from the diagnostic build (`make diag`)
      timeout -k 10 200 python tools/diag_ws.py "$@" | tee gpurun_out/diag_ws.txt ;;
    full)
      # the round's evidence pass: tests, smoke, kernel trace, traffic and SQ counters (exact + fma), bench
      local tag=${1:-r04}
      run_step test || return 1
      run_step smoke || return 1
      # (--no-other-configs: configs 4 and 5 launch kernels of the SAME name; the trace's per-kernel average must be config 3's alone)
      run_step prof ${tag}_bench --no-cpu-baseline --no-other-configs || return 1
      cp "$(find gpurun_out/prof_${tag}_bench -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_bench_kernel_stats.csv
      run_step traffic config3_exact_65536 "vs_synth_ws_kernel<0" || return 1
      run_step traffic config3_fma_65536 "vs_synth_ws_kernel<1" --arith fma || return 1
      run_step sq config3_exact_65536 "vs_synth_ws_kernel<0" || return 1
      run_step sq config3_fma_65536 "vs_synth_ws_kernel<1" --arith fma || return 1
      run_step bench || return 1
      cp gpurun_out/bench.json gpurun_out/${tag}_bench_n1.json ;;
    pmcset)
      # traffic + SQ passes for a set of keys on the tree as it stands -> profiles/pmc_{traffic,valu}.json (copy them back from gpurun_out/):
      #   tools/gpu.sh pmcset 3        config 3 in the three arithmetics + config 3 with vowel -n (fused kernel and noise pass)
      #   tools/gpu.sh pmcset 2 | 4 | 5     that configuration, exact and fma
      local which=${1:-3}
      if [ "$which" = 3 ]; then
        run_step traffic config3_exact_65536 "vs_synth_ws_kernel<0" || return 1
        run_step sq config3_exact_65536 "vs_synth_ws_kernel<0" || return 1
        run_step traffic config3_fma_65536 "vs_synth_ws_kernel<1" --arith fma || return 1
        run_step sq config3_fma_65536 "vs_synth_ws_kernel<1" --arith fma || return 1
        run_step traffic config3_f32_65536 "vs_synth_ws_kernel<2" --arith f32 || return 1
        run_step sq config3_f32_65536 "vs_synth_ws_kernel<2" --arith f32 || return 1
      elif [ "$which" = 3n ]; then
        run_step traffic config3_onoise_exact_65536 "vs_synth_ws_pow_kernel<0" --vowel-n 20 || return 1
        run_step sq config3_onoise_exact_65536 "vs_synth_ws_pow_kernel<0" --vowel-n 20 || return 1
        run_step traffic config3_onoise_noisepass_exact_65536 "vs_out_noise_kernel" --vowel-n 20 || return 1
        run_step sq config3_onoise_noisepass_exact_65536 "vs_out_noise_kernel" --vowel-n 20 || return 1
      elif [ "$which" = f ]; then
        # the rows of the bench line's other_configs that the sets above leave out: f32 on configs 2 / 4 / 5, vowel -n in fma and f32
        for c in 2 4 5; do
          local k=config$c; [ "$c" = 4 ] && k=config4_shard
          local n=65536; [ "$c" = 4 ] && n=32768; [ "$c" = 2 ] && n=1024
          run_step traffic ${k}_f32_$n "vs_synth_ws_kernel<2, true, 3" --config $c --arith f32 || return 1
          run_step sq ${k}_f32_$n "vs_synth_ws_kernel<2, true, 3" --config $c --arith f32 || return 1
        done
        run_step traffic config3_onoise_fma_65536 "vs_synth_ws_pow_kernel<1" --vowel-n 20 --arith fma || return 1
        run_step sq config3_onoise_fma_65536 "vs_synth_ws_pow_kernel<1" --vowel-n 20 --arith fma || return 1
        run_step traffic config3_onoise_f32_65536 "vs_synth_ws_pow_kernel<2" --vowel-n 20 --arith f32 || return 1
        run_step sq config3_onoise_f32_65536 "vs_synth_ws_pow_kernel<2" --vowel-n 20 --arith f32 || return 1
      else
        local key=config${which}; [ "$which" = 4 ] && key=config4_shard
        local lanes=65536; [ "$which" = 4 ] && lanes=32768; [ "$which" = 2 ] && lanes=1024
        local roles=3
        run_step traffic ${key}_exact_$lanes "vs_synth_ws_kernel<0, true, $roles" --config $which || return 1
        run_step sq ${key}_exact_$lanes "vs_synth_ws_kernel<0, true, $roles" --config $which || return 1
        run_step traffic ${key}_fma_$lanes "vs_synth_ws_kernel<1, true, $roles" --config $which --arith fma || return 1
        run_step sq ${key}_fma_$lanes "vs_synth_ws_kernel<1, true, $roles" --config $which --arith fma || return 1
      fi ;;
    configs)
      # the other BASELINE configurations under rocprofv3 --kernel-trace --stats:  tools/gpu.sh configs <tag>
      local tag=${1:-r04}
      for c in 2 4 5; do
        run_step prof ${tag}_config$c --config $c --no-cpu-baseline > /dev/null || return 1
        cp "$(find gpurun_out/prof_${tag}_config$c -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_config${c}_kernel_stats.csv
        cp gpurun_out/prof_${tag}_config$c.json gpurun_out/${tag}_config${c}_bench.json
      done ;;
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
