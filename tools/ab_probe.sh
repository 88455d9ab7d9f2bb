# A/B on one box: runs the given command alternately with the library of the working tree and with an older build kept in _exp_old/
# (git worktree add /tmp/oldwt <commit>; make -C /tmp/oldwt/otti_amd/csrc OBJDIR=/tmp/oldwt/bld OUT=$PWD/_exp_old $PWD/_exp_old/libottispartan.so).
# usage on the GPU box, from the repository root: bash tools/ab_probe.sh [reps] -- <command ...>
set -e
REPS=2; if [ "$2" = "--" ]; then REPS=$1; shift; fi; shift
cp otti_amd/libottispartan.so /tmp/lib_new.so
trap 'cp /tmp/lib_new.so otti_amd/libottispartan.so' EXIT
for rep in $(seq $REPS); do
  for v in new old; do
    if [ $v = new ]; then cp /tmp/lib_new.so otti_amd/libottispartan.so; else cp _exp_old/libottispartan.so otti_amd/libottispartan.so; fi
    echo "=== $v (rep $rep)"; "$@"
  done
done
