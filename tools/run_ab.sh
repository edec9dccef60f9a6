# A/B/C of several builds of the library on one box: usage  bash tools/run_ab.sh <tag> <command ...>
# every ab/lib_<name>.so is swapped in for the command in turn (two rounds); the tree's own build is restored at the end
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=$1; shift; O=gpurun_out/$T; mkdir -p $O
cp tomobar_amd/libtomo_mi355x.so /tmp/tree.so
for rep in 1 2; do
  for L in ab/lib_*.so; do
    n=$(basename $L .so); cp $L tomobar_amd/libtomo_mi355x.so
    "$@" > $O/${n}_$rep.txt 2>&1
  done
done
cp /tmp/tree.so tomobar_amd/libtomo_mi355x.so
for f in $O/*.txt; do echo "== $(basename $f .txt)"; grep -v amdgpu.ids $f; done
