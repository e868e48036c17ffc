cd "$GRAFT_REPO_ROOT"
for round in 1 2; do for lib in $LIBS; do
  echo "== $lib"
  PT_HIP_LIB=$PWD/tools/build/$lib timeout -k 10 120 python3 tools/tail_run.py 2>&1 | grep -v "^\[" || exit 1
done; done
