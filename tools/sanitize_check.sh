#!/bin/bash
# The test infrastructure's two native pieces under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only): the C oracle and the kernel
# sources compiled for the host (tests/emu).  Runs the oracle / emulation test files and two emulation campaigns (generous and stingy
# capacities) with both libraries instrumented; any report fails the script.  Usage: tools/sanitize_check.sh [programs per generator]
set -e
R=$(cd "$(dirname "$0")/.." && pwd); N=${1:-600}; T=$(mktemp -d /tmp/madsim_san.XXXX)
SAN="-O1 -g -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer"
gcc -std=c11 -pthread $SAN -o $T/liboracle_san.so $R/oracle/madsim_oracle.c
g++ -std=c++17 -DMADSIM_EMU $SAN -x c++ -I$R/tests/emu -o $T/libemu_san.so $R/tests/emu/emu_driver.cpp
export LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 MADSIM_ORACLE_LIB=$T/liboracle_san.so MADSIM_EMU_LIB=$T/libemu_san.so
cd $R
python -m pytest tests/test_oracle_kat.py tests/test_oracle_lifecycle.py tests/test_oracle_properties.py tests/test_emu_parity.py -x -q 2>&1 | tee $T/pytest.log | tail -n 2
python tools/emu_campaign.py $N 97000000 2>&1 | tee $T/c1.log | tail -n 1
python tools/emu_campaign.py $N 97500000 tight 2>&1 | tee $T/c2.log | tail -n 1
if grep -q "runtime error\|AddressSanitizer" $T/*.log; then echo "SANITIZER REPORTS:"; grep "runtime error\|AddressSanitizer" $T/*.log | sort | uniq -c | head; exit 1; fi
echo "sanitize check ok: no ASan / UBSan report"
