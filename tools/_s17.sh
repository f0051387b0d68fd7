cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in main mix1 mix4; do so=""; [ $v != main ] && so=$PWD/imagepipe_amd/csrc/build/ablate/lib$v.so; echo "== $v"; IPK_SO_OVERRIDE=$so python tools/mix_probe.py 2>/dev/null; done
