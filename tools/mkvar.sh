#!/bin/bash
# mkvar.sh <source basename (no .hip)> <out.so> [extra hipcc flags...]: library with ONE source recompiled with extra flags
src=$1; out=$2; shift 2
cd /root/repo
FILEFLAGS=""
case $src in gemm_bf16|gemm8p_bf16|attention_fwd|attention_bwd|skinny) FILEFLAGS="-mllvm -amdgpu-mfma-vgpr-form=1";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function -ffp-contract=off $FILEFLAGS "$@" -c gaviko_amd/csrc/$src.hip -o /tmp/_var_$$.o || exit 1
objs=""
for f in gaviko_amd/csrc/*.hip; do b=$(basename $f .hip); if [ "$b" != "$src" ]; then objs="$objs gaviko_amd/csrc/_obj/$b.o"; fi; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out $objs /tmp/_var_$$.o && rm -f /tmp/_var_$$.o && echo built $out
