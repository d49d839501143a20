"""Extracts the reference's own example outputs (MetaMaps_example_output.zip, real miniSeq+H run of the
reference: 100 HMP reads, k=16 w=16) into tests/golden/example/.  These are DATA files shipped with the
reference (format + known-answer fixtures), not source.  The 54k-line contigCoverage file is skipped.
Run in the build container only (needs /root/reference)."""
import os
import zipfile

SRC = "/root/reference/MetaMaps_example_output.zip"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "example")
KEEP = (".EM", ".EM.WIMP", ".EM.reads2Taxon", ".EM.reads2Taxon.krona", ".meta", ".meta.unmappedReadsLengths", ".parameters",
        ".EM.lengthAndIdentitiesPerMappingUnit", ".EM.evidenceUnknownSpecies")
os.makedirs(DST, exist_ok=True)
with zipfile.ZipFile(SRC) as z:
    for n in z.namelist():
        base = os.path.basename(n)
        if not base:
            continue
        suffix = base[len("hmp7_2_short_miniSeq+H"):]
        if suffix in KEEP:
            with open(os.path.join(DST, "example" + suffix), "wb") as f:
                f.write(z.read(n))
            print("wrote", "example" + suffix)
