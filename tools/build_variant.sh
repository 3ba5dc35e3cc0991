#!/bin/bash
# build a variant of libihm2mpc.so with extra compiler flags into build/<name>/ (diagnostics only; never the product)
# usage: tools/build_variant.sh name "-DFLAG ..."
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
dst=$root/build/$name
rm -rf $dst; mkdir -p $dst
cp $root/ihm2_amd/csrc/*.hip $root/ihm2_amd/csrc/*.hpp $root/ihm2_amd/csrc/*.h $root/ihm2_amd/csrc/Makefile $dst/
sed -i "s|^CXXFLAGS = \(.*\)$|CXXFLAGS = \1 $*|; s|^OUT     = .*|OUT     = libihm2mpc_var.so|; s|\.\./\.\./include|$root/include|g" $dst/Makefile
sed -i "s|\"../../include/ihm2mpc.h\"|\"$root/include/ihm2mpc.h\"|" $dst/ihm2mpc_internal.h
make -C $dst -j8 -s 2>&1 | grep -E "error|Error" || true
ls -la $dst/libihm2mpc_var.so
