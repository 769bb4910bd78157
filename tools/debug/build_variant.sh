#!/bin/bash
# Debug aid: another build of libsshash_amd.so with extra -D flags on engine.hip only, next to the real one -- compiled by hipcc in one
# go, i.e. WITHOUT tools/isa_guard.py (csrc/Makefile): `build_variant.sh unguarded` is the library with the gfx950 shift hazard in it.
#   tools/debug/build_variant.sh unguarded                                  ->  tools/debug/libsshash_amd_unguarded.so
#   tools/debug/build_variant.sh codes -DSSHASH_DEBUG_MEMBER_CODES_DEFERRED   (is_member stores what each pass thought: member_race.py)
# Use it with SSHASH_AMD_LIBRARY=<that file> (sshash_amd/_binding.py). The other objects come from the regular build.
set -e
cd "$(dirname "$0")/../../sshash_amd/csrc"
name=$1; shift
make -j8 >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c engine.hip -o /tmp/engine_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC index.o reads.o capi.o /tmp/engine_$name.o streaming.o sktable.o sharded.o -lz -lpthread -ldl \
  -o ../../tools/debug/libsshash_amd_$name.so
echo built tools/debug/libsshash_amd_$name.so
