#!/bin/bash
# pmc_pass <outdir> <counter> [<counter> ...] -- <program> [args]
#
# ONE rocprofv3 counter pass, the only way this repository collects PMC counters on the GPU box.  It exists so
# that the rules of MI355X_MICROARCH.md's HBM / rocprofv3 section cannot be forgotten in an ad-hoc command line:
#   * the derived TCC byte counters (FETCH_SIZE, WRITE_SIZE, and the TCC_EA0_* request counters they are built
#     from) go ONE PER PASS -- two of them in one pass is "Request exceeds the capabilities of the hardware",
#     signal 6, and a process that sits until its timeout (round 3, gpurun_out/call105.log);
#   * counters only with --kernel-trace: never with --sys-trace / --runtime-trace / the hip / hsa / memory-copy
#     domains (gpurun refuses that combination);
#   * the program itself after `--` (no env / bash -c / taskset hop: the profiler's library has initialised the GPU
#     before the program starts, so every such hop is a forbidden exec);
#   * a bounded run: timeout -k 10.
# Source it (`. tools/pmc_pass.sh`) or call it.
pmc_pass() {
  local out=$1; shift
  local counters=() heavy=0
  while [ $# -gt 0 ] && [ "$1" != "--" ]; do
    case "$1" in
      FETCH_SIZE|WRITE_SIZE|TCC_EA0_RDREQ*|TCC_EA0_WRREQ*|TCC_EA_RDREQ*|TCC_EA_WRREQ*) heavy=$((heavy + 1));;
      -*) echo "pmc_pass: options are not accepted here ($1): counters, then --, then the program" >&2; return 2;;
    esac
    counters+=("$1"); shift
  done
  [ "${1:-}" = "--" ] || { echo "pmc_pass: missing -- before the program" >&2; return 2; }
  shift
  [ ${#counters[@]} -gt 0 ] || { echo "pmc_pass: no counters" >&2; return 2; }
  if [ $heavy -gt 1 ]; then
    echo "pmc_pass: ${counters[*]}: the TCC byte / request counters go one per pass (profiles/README.md)" >&2
    return 2
  fi
  case "$(basename "$1")" in
    env|bash|sh|taskset|numactl|timeout) echo "pmc_pass: put the program itself after -- (not $1)" >&2; return 2;;
  esac
  mkdir -p "$out"
  ( cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}" &&
    timeout -k 10 "${PMC_PASS_TIMEOUT:-300}" rocprofv3 --kernel-trace --pmc "${counters[@]}" --output-format csv \
      -d "$out" -- "$@" > "$out.log" 2>&1 )
}
if [ "${BASH_SOURCE[0]}" = "$0" ]; then
  pmc_pass "$@"
fi
