cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in main demo8; do
  so=""; [ $v != main ] && so=$PWD/imagepipe_amd/csrc/build/ablate/lib$v.so
  echo -n "$v: "; IPK_SO_OVERRIDE=$so python tools/stage_probe.py demosaic 2>/dev/null | tail -1
done; done
