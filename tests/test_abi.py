"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU, exports every entry point
include/bwahip.h declares, and its structs are layout-compatible with the reference's (checked against the
reference headers when they are present).  No compute calls."""
import ctypes as C
import os
import re
import subprocess
import pytest
import common
from common import bw

HDR = os.path.join(common.ROOT, "include", "bwahip.h")


def test_library_exports_every_declared_symbol(built):
    text = open(HDR).read()
    declared = sorted(set(re.findall(r"\b(bwahip_[a-z0-9_]+)\s*\(", text)))
    assert len(declared) >= 20
    lib = C.CDLL(bw.LIB_PATH)
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, f"declared in bwahip.h but not exported: {missing}"


def test_compat_library_exports_mem_process_seqs(built):
    """libbwamem_hip.so exports the symbol the north star names, mem_process_seqs (bwamem.h:69), and the two helpers
    include/bwamem_hip.h declares; it resolves against libbwahip.so."""
    lib = C.CDLL(os.path.join(os.path.dirname(bw.LIB_PATH), "libbwamem_hip.so"))
    for sym in ("mem_process_seqs", "bwahip_compat_set_rg_id", "bwahip_compat_release"):
        assert hasattr(lib, sym), sym
    assert os.access(os.path.join(common.ROOT, "tests", "c_abi_driver"), os.X_OK), "plain-C driver was not built"


@pytest.mark.skipif(not os.path.exists("/root/reference/bwamem.h"), reason="reference headers not present")
def test_reference_translation_unit_links_against_compat_library(built, tmp_path):
    """A C file that includes only the REFERENCE's bwamem.h and calls mem_process_seqs through the reference's prototype
    compiles and links against libbwamem_hip.so with no bwamem.o in sight (never run here: no GPU)."""
    src = tmp_path / "caller.c"
    src.write_text(r'''
#include "bwamem.h"
int main(int argc, char **argv) {
    mem_opt_t *opt = 0; bwt_t *bwt = 0; bntseq_t *bns = 0; uint8_t *pac = 0; bseq1_t *seqs = 0;
    if (argc > 100) mem_process_seqs(opt, bwt, bns, pac, 0, 0, seqs, 0);   /* bwamem.h:69 */
    return 0; }''')
    exe = tmp_path / "caller"
    pkg = os.path.dirname(bw.LIB_PATH)
    subprocess.check_call(["gcc", "-I/root/reference", str(src), "-o", str(exe), "-L" + pkg, "-lbwamem_hip", "-lbwahip", "-Wl,-rpath," + pkg])
    r = subprocess.run(["nm", "-D", "--undefined-only", str(exe)], stdout=subprocess.PIPE, text=True)
    assert "mem_process_seqs" in r.stdout


def test_struct_sizes_match_reference_layouts(built):
    assert C.sizeof(bw.Opt) == 168 and bw.Opt.mat.offset == 136          # mem_opt_t (bwa.h:86-118)
    assert C.sizeof(bw.AlnReg) == 88                                        # mem_alnreg_t (bwa.h:145-163)
    assert C.sizeof(bw.Seq) == 56                                           # bseq1_t (bwa.h:58-63)
    assert C.sizeof(bw.PeStat) == 32 and C.sizeof(bw.Bwt) == 1120 and C.sizeof(bw.Ann) == 40 and C.sizeof(bw.Bns) == 48


def test_default_options_match_mem_opt_init(built):
    o = bw.default_opt()
    assert (o.a, o.b, o.o_del, o.e_del, o.o_ins, o.e_ins, o.w, o.T, o.zdrop) == (1, 4, 6, 1, 6, 1, 100, 30, 100)
    assert (o.min_seed_len, o.split_width, o.max_occ, o.max_mem_intv, o.max_chain_gap) == (19, 10, 500, 20, 10000)
    assert o.mapQ_coef_fac == 3 and o.chunk_size == 30000000 and o.max_matesw == 50
    assert list(o.mat) == [1, -4, -4, -4, -1, -4, 1, -4, -4, -1, -4, -4, 1, -4, -1, -4, -4, -4, 1, -1, -1, -1, -1, -1, -1]


def test_no_gpu_is_a_loud_error_not_a_fallback(built, tmp_path):
    """Without a usable device the hot path must refuse to run (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(bw.BwahipError):
        bw.Context(str(tmp_path / "nonexistent"))


@pytest.mark.skipif(not os.path.exists("/root/reference/bwamem.h"), reason="reference headers not present")
def test_mirrors_are_layout_identical_to_reference_headers(built, tmp_path):
    src = tmp_path / "chk.c"
    src.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "bwamem.h"
#include "bwahip.h"
#define CHK(a,b) if (sizeof(a) != sizeof(b)) { printf("size " #a "\n"); bad = 1; }
#define OFF(a,b,f) if (offsetof(a,f) != offsetof(b,f)) { printf("offset " #a "." #f "\n"); bad = 1; }
int main(void) { int bad = 0;
 CHK(bwt_t,bwahip_bwt_t) CHK(bntann1_t,bwahip_ann_t) CHK(bntamb1_t,bwahip_amb_t) CHK(bntseq_t,bwahip_bns_t) CHK(bseq1_t,bwahip_seq_t)
 CHK(mem_opt_t,bwahip_opt_t) CHK(mem_alnreg_t,bwahip_alnreg_t) CHK(mem_alnreg_v,bwahip_alnreg_v) CHK(mem_pestat_t,bwahip_pestat_t) CHK(bwtintv_t,bwahip_intv_t)
 OFF(mem_opt_t,bwahip_opt_t,mat) OFF(mem_opt_t,bwahip_opt_t,max_occ) OFF(mem_opt_t,bwahip_opt_t,mask_level) OFF(mem_opt_t,bwahip_opt_t,mapQ_coef_fac)
 OFF(bwt_t,bwahip_bwt_t,sa) OFF(bwt_t,bwahip_bwt_t,sa_intv) OFF(bwt_t,bwahip_bwt_t,bwt)
 OFF(mem_alnreg_t,bwahip_alnreg_t,seedlen0) OFF(mem_alnreg_t,bwahip_alnreg_t,frac_rep) OFF(mem_alnreg_t,bwahip_alnreg_t,secondary)
 OFF(bseq1_t,bwahip_seq_t,sam) OFF(bntseq_t,bwahip_bns_t,anns) OFF(bntann1_t,bwahip_ann_t,is_alt)
 return bad; }''')
    exe = tmp_path / "chk"
    subprocess.check_call(["gcc", "-I/root/reference", "-I" + os.path.join(common.ROOT, "include"), str(src), "-o", str(exe)])
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stdout
