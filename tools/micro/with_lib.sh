#!/bin/bash
# run a command with another build of the library in place (the GPU box works on a copy of the tree): tools/micro/with_lib.sh <lib.so> <command ...>
lib=$1; shift
cp smcp_amd/libsmcp_amd.so /tmp/orig_lib.so
cp "$lib" smcp_amd/libsmcp_amd.so
"$@"; rc=$?
cp /tmp/orig_lib.so smcp_amd/libsmcp_amd.so
exit $rc
