"""CPU: the host side of libgcn_hip.so under AddressSanitizer and UndefinedBehaviorSanitizer.

`make -C gcn_amd/csrc asan` compiles the library's sources and gcn_amd/csrc/asan_check.cpp into one
executable with the sanitizers on the host code and runs it: the size queries, the argument validation of
every entry point of the C ABI (plan creation and the RCCL exchange included) and the host half of plan
creation on random row lengths.  No GPU is needed; objects are kept under gcn_amd/csrc/asan/ between runs."""
import os
import subprocess

from conftest import ROOT


def test_host_code_is_clean_under_asan_and_ubsan():
    csrc = os.path.join(ROOT, 'gcn_amd', 'csrc')
    run = subprocess.run(['make', '-C', csrc, '-j8', 'asan'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         text=True, timeout=900)
    assert run.returncode == 0, run.stdout[-4000:]
    assert 'asan_check ok' in run.stdout
    assert 'runtime error' not in run.stdout and 'AddressSanitizer' not in run.stdout
