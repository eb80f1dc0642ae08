"""The Swift package under swift/ cannot be compiled in the build image (no Swift toolchain), so it is held to its two
contracts by reading the sources:
  * the C ABI: every `he_*` call names a function include/he_amd.h declares and passes as many arguments as it takes;
  * the reference's protocol: `GpuBfv` declares every static member a HeScheme conformer must supply itself -- the list
    NoOpScheme supplies (reference NoOpScheme.swift:35-368) -- and both twins of every requirement it accelerates
    (HeScheme.swift:190-1090), `GpuPirUtil` the three PirUtilProtocol requirements of PirUtil.swift:38-147 it answers;
  * every Swift listing is an excerpt of a file under swift/.
"""
import glob
import json
import os
import re

import pytest

from swift_source import c_calls, header_functions, static_members, strip_comments

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SWIFT = os.path.join(ROOT, "swift")
SOURCES = sorted(glob.glob(os.path.join(SWIFT, "Sources", "HeAmd", "*.swift")) +
                 glob.glob(os.path.join(SWIFT, "Tests", "**", "*.swift"), recursive=True))
GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "noop_scheme_members.json")))


def read(path):
    return open(path).read()


def test_the_shipped_header_is_the_product_header():
    assert read(os.path.join(SWIFT, "Sources", "CHeAmd", "include", "he_amd.h")) == read(
        os.path.join(ROOT, "include", "he_amd.h"))


def test_every_c_call_matches_the_header():
    header = read(os.path.join(ROOT, "include", "he_amd.h"))
    declared = header_functions(header)
    assert len(declared) > 100
    types = set(re.findall(r"typedef[^;]*?\b(he_[a-z_]+)\s*;", header)) | set(re.findall(r"\}\s*(he_[a-z_]+)\s*;", header)) | set(
        re.findall(r"\b(?:enum|struct)\s+(he_[a-z_]+)", header))
    seen = set()
    for path in SOURCES:
        for name, argument_count in c_calls(read(path)):
            if name in types:  # `he_status(UInt32(code))`: a C type's initialiser, not a call
                continue
            assert name in declared, f"{os.path.relpath(path, ROOT)} calls {name}, which include/he_amd.h does not declare"
            assert argument_count == declared[name], (
                f"{os.path.relpath(path, ROOT)}: {name} takes {declared[name]} arguments, called with {argument_count}")
            seen.add(name)
    # the surface the package is there to reach
    for name in ["he_bfv_mul_device", "he_bfv_relinearize_device", "he_bfv_apply_galois_device",
                 "he_bfv_mod_switch_down_device", "he_bfv_mod_switch_down_to_single_device", "he_bfv_mul_plain_device",
                 "he_bfv_inner_product_device", "he_bfv_inner_product_plain_device", "he_bfv_add_plain_device",
                 "he_bfv_sub_plain_device", "he_ntt_forward_device", "he_ntt_inverse_device",
                 "he_pir_compute_response_to_query_device", "he_pir_expand_device",
                 "he_pir_compute_response_chunk_device", "he_stream_add_callback", "he_get_device",
                 "he_set_scratch_cache"]:
        assert name in seen, f"no Swift file calls {name}"


def gpu_bfv_members():
    text = "".join(read(p) for p in SOURCES if os.path.basename(p).startswith("GpuBfv"))
    return static_members(text, "GpuBfv")


def test_gpu_bfv_supplies_what_a_conformer_must():
    members = gpu_bfv_members()
    missing = [m for m in GOLDEN["noop_scheme_static_members"] if m not in members]
    assert not missing, f"GpuBfv lacks members NoOpScheme declares: {missing}"


def test_gpu_bfv_overrides_both_twins_of_what_it_accelerates():
    members = gpu_bfv_members()
    missing = [m for m in GOLDEN["accelerated_protocol_requirements"] if m not in members]
    assert not missing, f"GpuBfv leaves accelerated requirements to the blocking defaults: {missing}"
    # every accelerated member reaches the device through GpuBfv+Device.swift: its helpers call exactly one entry point
    device = read(os.path.join(SWIFT, "Sources", "HeAmd", "GpuBfv+Device.swift"))
    for helper in ["translate", "multiplyPlain", "multiply", "relinearization", "galois", "modulusSwitch", "transform",
                   "ciphertextInnerProduct", "plaintextInnerProduct"]:
        assert re.search(r"static func %s\b" % helper, device), helper
        assert re.search(r"try %s\(" % helper, read(os.path.join(SWIFT, "Sources", "HeAmd", "GpuBfv.swift"))), helper


@pytest.mark.skipif(not os.path.isdir("/root/reference/Sources/HomomorphicEncryption"),
                    reason="the reference checkout exists in the build container only")
def test_the_golden_member_list_is_the_reference_s():
    reference = "/root/reference/Sources/HomomorphicEncryption"
    noop = static_members(read(os.path.join(reference, "NoOpScheme.swift")), "NoOpScheme")
    assert sorted(noop) == GOLDEN["noop_scheme_static_members"]
    protocol = static_members(read(os.path.join(reference, "HeScheme.swift")), "HeScheme", kind="protocol")
    for requirement in GOLDEN["accelerated_protocol_requirements"]:
        assert requirement in protocol, requirement


def test_gpu_pir_util_answers_the_server_side_requirements():
    text = read(os.path.join(SWIFT, "Sources", "HeAmd", "GpuPirUtil.swift"))
    assert re.search(r"public enum GpuPirUtil<Scheme: HeScheme>: PirUtilProtocol", text)
    members = static_members(text, "GpuPirUtil")
    for requirement in ["computeResponse(to:using:databases:parameter:context:callOptions:)",
                        "expand(ciphertexts:outputCount:using:callOptions:)",
                        "computeResponseForOneChunk(expandedDim0Query:expandedRemainingQuery:dataChunk:using:parameter:callOptions:)"]:
        assert requirement in members, (requirement, sorted(members))
    # the whole-query path must not upload the database: it goes through the resident cache
    body = strip_comments(text)
    whole_query = body[body.index("func computeResponse("):body.index("func expand(")]
    assert "GpuResidentCache.shared.resident" in whole_query and "database.upload" not in whole_query


def swift_blocks(markdown):
    return re.findall(r"```swift\n(.*?)```", markdown, flags=re.S)


def test_integration_md_listings_are_excerpts_of_the_package():
    markdown = read(os.path.join(ROOT, "INTEGRATION.md"))
    blocks = swift_blocks(markdown)
    assert len(blocks) >= 5
    corpus = "\n".join(read(p) for p in glob.glob(os.path.join(SWIFT, "**", "*.swift"), recursive=True))
    normalised = re.sub(r"\s+", " ", corpus)
    for block in blocks:
        for piece in block.split("\n// ...\n"):  # an elision marker on its own line splits an excerpt
            piece = re.sub(r"\s+", " ", piece).strip()
            assert piece and piece in normalised, f"INTEGRATION.md shows Swift that is not in swift/:\n{piece[:300]}"


def test_ci_recipe_names_what_is_missing():
    """swift/ci.sh -- the one-command build + test of the package against a reference checkout -- refuses to start without
    its prerequisites and says which ones (there is no Swift toolchain in this image: exit status 2, nothing built)."""
    import shutil
    import subprocess

    script = os.path.join(SWIFT, "ci.sh")
    assert os.access(script, os.X_OK)
    if shutil.which("swift") is not None:
        pytest.skip("a Swift toolchain is present: run swift/ci.sh itself")
    result = subprocess.run(["bash", script, "/nonexistent/checkout"], capture_output=True, text=True)
    assert result.returncode == 2
    assert "no swift on PATH" in result.stderr and "no reference checkout at /nonexistent/checkout" in result.stderr
