cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SO=$PWD/imagepipe_amd/csrc/build/ablate/libsweepadj.so
for m in 1 2 3 4 6; do echo -n "demosaic mul=$m: "; IPK_SO_OVERRIDE=$SO IPK_DEMO_MUL=$m python tools/stage_probe.py demosaic 2>/dev/null | tail -1; done
for m in 8 12 16 24; do echo -n "gamma adj mul=$m: "; IPK_SO_OVERRIDE=$SO IPK_GAMMA_MUL=$m python tools/stage_probe.py gamma 2>/dev/null | tail -1; done
for m in 6 8 12 16; do echo -n "tolab mul=$m: "; IPK_SO_OVERRIDE=$SO IPK_TOLAB_MUL=$m python tools/stage_probe.py tolab 2>/dev/null | tail -1; done
