#!/bin/bash
# HBM traffic of the dominant kernel from the PMC counters, as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (they do not fit one pass),
# no other trace domain alongside.  Usage: tools/pmc_traffic.sh <outdir> -- <command ...>
out=$1; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -- "$@" > $out/$c.log 2>&1 || echo "$c pass failed"
done
