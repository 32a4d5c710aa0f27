"""The drop-in boundary, compiled and run: integration/align_gpu.c is the file a maintainer of viq854/bwbble adds to
mg-aligner/ (INTEGRATION.md).  Here it is compiled against the REFERENCE's headers, linked with the REFERENCE's own objects
(oracle/Makefile `dropin`; build container only - /root/reference does not exist on the GPU box, the built binary travels
in oracle/_ref/) and, on the GPU, run on the reference's own bwt_t / reads_t against the reference's golden .aln files."""
import os
import shutil
import subprocess

import pytest

from golden.make_golden import ALIGN_CONFIGS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "oracle", "_ref", "bwbble_dropin")
REF_SRC = "/root/reference/mg-aligner"


@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="reference sources only exist in the build container")
def test_binding_compiles_and_links_against_the_reference(built):
    """-I<reference>/mg-aligner: the _Static_asserts in the binding prove bwb_params == aln_params_t field for field
    (align.h:48-79), the link proves every reference symbol it uses (add_alignment, alns2alnf_bin, init_alignments, ...)
    and every C-ABI symbol resolves."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    assert os.path.exists(DROPIN)
    syms = subprocess.run(["nm", DROPIN], check=True, stdout=subprocess.PIPE, text=True).stdout
    for s in ("T align_reads_inexact_gpu", "T add_alignment", "T alns2alnf_bin", "T fastq2reads", "T load_bwt", "U bwb_hip_slot_submit", "U bwb_hip_ctx_create"):
        assert s in syms, s
    # a params struct that drifts from the reference's must not compile
    bad = subprocess.run(["gcc", "-c", "-std=gnu11", "-fopenmp", "-I" + REF_SRC, "-I" + os.path.join(ROOT, "include"), "-Dmax_gapo=max_gapo, extra_field",
                          os.path.join(ROOT, "integration", "align_gpu.c"), "-o", os.devnull], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert bad.returncode != 0


@pytest.mark.gpu
@pytest.mark.parametrize("fq,name", [("toy.fq", "n3"), ("toy.fq", "n0"), ("ragged.fq", "n4gap"), ("ragged.fq", "s2"), ("toy.fq", "p2")])
def test_reference_types_through_the_binding_give_reference_bytes(golden, tmp_path, fq, name):
    if not os.path.exists(DROPIN):
        pytest.skip("oracle/_ref/bwbble_dropin is built in the build container (make -C oracle) and travels with gpurun")
    for ext in ("", ".bwt", ".ann"):
        shutil.copy(os.path.join(golden, "toy.fa" + ext), tmp_path / ("toy.fa" + ext))
    out = tmp_path / "o.aln"
    subprocess.run([DROPIN, str(tmp_path / "toy.fa"), os.path.join(golden, fq), str(out)] + ALIGN_CONFIGS[name], check=True, stdout=subprocess.DEVNULL)
    tag = "toy" if fq == "toy.fq" else "ragged"
    assert open(out, "rb").read() == open(os.path.join(golden, f"{tag}_{name}.aln"), "rb").read()
