# usage: bash tools/repro_check.sh [bench args]  -- runs the bench twice and compares final KL energy and work counters:
# every sum on the path is built in a fixed order or in fixed point, so the two lines must agree to the last digit
cd "${GRAFT_REPO_ROOT:?}"
run() { timeout 900 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(repr(d['final_kl_energy']), d['per_step_counts_rank0'])"; }
a=$(run "$@"); b=$(run "$@")
echo "run 1: $a"; echo "run 2: $b"
[ "$a" = "$b" ] && echo "REPRODUCIBLE" || { echo "DIFFERENT"; exit 1; }
