"""The C-ABI library loads and exports every symbol include/*.h declares;
struct layouts seen from ctypes match the header (sizes computed by gcc);
and there is no CPU fallback behind the compute entry points."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

from conftest import ROOT, has_gpu


def declared_functions():
    names = set()
    for h in ("tbcheck.h", "tbsynth.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        for m in re.finditer(r"\b(tb[cs]_[a-z0-9_]+)\s*\(", src):
            names.add(m.group(1))
    names.discard("tbc_step_fn")
    return sorted(names)


def test_every_declared_symbol_is_exported_and_bound(native):
    lib = native.lib()
    decl = declared_functions()
    assert len(decl) >= 15
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
        assert name in native.SYMBOLS, f"{name} has no ctypes binding"
    assert lib.tbc_version() == 2
    assert lib.tbc_strerror(3).decode().startswith("no usable gfx950")


def test_struct_layouts_match_header(native):
    prog = r'''
#include <stdio.h>
#include "tbcheck.h"
#include "tbsynth.h"
int main(void){ printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(tbc_events), sizeof(tbc_ops),
  sizeof(tbc_model), sizeof(tbc_opts), sizeof(tbc_config), sizeof(tbc_counters), sizeof(tbc_result),
  sizeof(tbc_batch_desc), sizeof(tbs_params), offsetof(tbc_result, counters),
  sizeof(tbc_batch_input), offsetof(tbc_batch_input, word), sizeof(tbc_input_info), offsetof(tbc_input_info, ops_cap),
  sizeof(tbc_progress), offsetof(tbc_progress, elapsed_ns)); return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "s.c")
        open(src, "w").write(prog)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    N = native
    mine = [C.sizeof(N.Events), C.sizeof(N.Ops), C.sizeof(N.Model), C.sizeof(N.Opts), C.sizeof(N.Config),
            C.sizeof(N.Counters), C.sizeof(N.Result), C.sizeof(N.BatchDesc), C.sizeof(N.SynthParams),
            N.Result.counters.offset,
            C.sizeof(N.BatchInput), N.BatchInput.word.offset, C.sizeof(N.InputInfo), N.InputInfo.ops_cap.offset,
            C.sizeof(N.Progress), N.Progress.elapsed_ns.offset]
    assert mine == sizes
    # the wire word of the streaming inputs, as the header's macro packs it
    assert N.WIRE_NIL == 0xFF and N.COMM_ID_BYTES == 128


def test_result_offsets_the_jna_shim_reads(native):
    """INTEGRATION.md's Clojure shim reads tbc_result by byte offset; tbc_result.search_width took the padding hole in
    front of the witness pointer this round -- nothing else may have moved."""
    prog = r'''
#include <stdio.h>
#include "tbcheck.h"
int main(void){ printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", offsetof(tbc_result, valid), offsetof(tbc_result, cause),
  offsetof(tbc_result, analyzer), offsetof(tbc_result, fail_op), offsetof(tbc_result, prev_ok_op), offsetof(tbc_result, final_state),
  offsetof(tbc_result, n_witness), offsetof(tbc_result, search_width), offsetof(tbc_result, witness), offsetof(tbc_result, n_configs),
  offsetof(tbc_result, configs)); return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "o.c")
        open(src, "w").write(prog)
        exe = os.path.join(d, "o")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        offs = [int(x) for x in subprocess.check_output([exe]).split()]
    assert offs == [0, 4, 8, 12, 16, 20, 24, 28, 32, 40, 44]
    R = native.Result
    assert [R.valid.offset, R.fail_op.offset, R.search_width.offset, R.witness.offset, R.n_configs.offset, R.configs.offset] == [0, 12, 28, 32, 40, 44]
    assert C.sizeof(R) == 960


def test_setfull_and_opts_offsets_the_jna_shim_writes(native):
    """INTEGRATION.md fills tbc_setfull_in / tbc_setfull_out and tbc_opts by byte offset (the set-full binding, lanes_per_history)."""
    prog = r'''
#include <stdio.h>
#include "tbcheck.h"
int main(void){ printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu  %zu %zu %zu %zu %zu %zu %zu  %zu %zu %zu\n",
  sizeof(tbc_setfull_in), offsetof(tbc_setfull_in, n_elements), offsetof(tbc_setfull_in, n_reads), offsetof(tbc_setfull_in, words_per_row),
  offsetof(tbc_setfull_in, device), offsetof(tbc_setfull_in, add_invoke), offsetof(tbc_setfull_in, add_ok), offsetof(tbc_setfull_in, read_invoke),
  offsetof(tbc_setfull_in, read_ok), offsetof(tbc_setfull_in, present),
  sizeof(tbc_setfull_out), offsetof(tbc_setfull_out, known), offsetof(tbc_setfull_out, last_present), offsetof(tbc_setfull_out, last_absent),
  offsetof(tbc_setfull_out, ns_scan), offsetof(tbc_setfull_out, bytes_scanned), offsetof(tbc_setfull_out, bytes_matrix),
  sizeof(tbc_opts), offsetof(tbc_opts, lanes_per_history), offsetof(tbc_opts, dominance)); return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "o.c")
        open(src, "w").write(prog)
        exe = os.path.join(d, "o")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        offs = [int(x) for x in subprocess.check_output([exe]).split()]
    assert offs == [56, 0, 4, 8, 12, 16, 24, 32, 40, 48, 48, 0, 8, 16, 24, 32, 40, 64, 56, 52]
    assert C.sizeof(native.SetFullIn) == 56 and C.sizeof(native.SetFullOut) == 48 and C.sizeof(native.Opts) == 64
    assert native.Opts.lanes_per_history.offset == 56
    # tbc_setfull_rows (compact reads, ABI version 2): 72 bytes, pointers from offset 16
    assert C.sizeof(native.SetFullRows) == 72 and native.SetFullRows.top.offset == 48 and native.SetFullRows.exc.offset == 64


@pytest.mark.skipif(has_gpu(), reason="only meaningful on a box without a GPU")
def test_no_cpu_fallback(native):
    from jepsen_tigerbeetle_amd import columns, core, synth
    assert native.lib().tbc_device_count() == 0
    ops = columns.pair_events(synth.register_events(n_ops=20, n_procs=3, seed=1))
    with pytest.raises(native.NoDeviceError):
        core.check_ops(ops, core.make_model(native.MODEL_CAS_REGISTER, native.NIL))
    with pytest.raises(native.NoDeviceError):
        core.Batch([ops], core.make_model(native.MODEL_CAS_REGISTER, native.NIL))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "jepsen-tigerbeetle_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M), f
                assert "liboracle" not in txt and "oracle.wgl" not in txt, f   # comments may cite oracle/*.c


def test_comm_objects_need_no_device(native):
    """tbc_comm_init_host only records the caller's transport: it can be made, asked and destroyed on a machine without a GPU;
    bad ranks are refused; the RCCL transport says why it is not there instead of crashing when librccl or the device is missing."""
    lib = native.lib()
    calls = []

    @native.ALLGATHER_FN
    def gather(user, send, recv, nbytes):
        calls.append(nbytes)
        return 0

    h = C.c_void_p()
    assert lib.tbc_comm_init_host(1, 2, gather, None, C.byref(h)) == 0
    assert (lib.tbc_comm_rank(h), lib.tbc_comm_world(h)) == (1, 2)
    lib.tbc_comm_destroy(h)
    assert lib.tbc_comm_init_host(2, 2, gather, None, C.byref(h)) == native.ERR_INVALID_ARG
    assert lib.tbc_batch_sweep_allgather(None, None, None) == native.ERR_INVALID_ARG
    if not has_gpu():
        ident = (C.c_uint8 * 128)()
        st = lib.tbc_comm_init(0, 1, ident, 0, C.byref(h))
        assert st in (native.ERR_NO_DEVICE, native.ERR_UNSUPPORTED), st
