#!/bin/bash
# After a GPU memory fault the runtime leaves gpucore.<pid> in the working directory: print which kernel's waves were at fault
# (rocgdb reads AMDGPU core files).  usage: tools/gpucore_report.sh <out-file> -- <command ...>
out=$1; shift; [ "$1" = "--" ] && shift
mkdir -p "$(dirname "$out")"
rm -f gpucore.*
"$@"; rc=$?
core=$(ls -t gpucore.* 2>/dev/null | head -n 1)
if [ -n "$core" ]; then
    ls -l "$core" > "$out"
    timeout -k 10 300 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "info agents" -ex "info dispatches" -ex "info threads" "$(command -v python3)" "$core" > "$out.full" 2>&1
    grep -v "^\[New" "$out.full" | cut -c1-260 | awk 'NR<=400' >> "$out"
    # the waves that stopped on a memory violation, their pc and the instructions around it
    timeout -k 10 300 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "thread apply all -q -s x/6i \$pc-16" "$(command -v python3)" "$core" 2>&1 | cut -c1-200 | awk 'NR<=300' > "$out.pc"
    rm -f "$core"
fi
exit $rc
