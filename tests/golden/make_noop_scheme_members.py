"""Writes tests/golden/noop_scheme_members.json: what a HeScheme conformer must supply itself = the static members
NoOpScheme declares (reference Sources/HomomorphicEncryption/NoOpScheme.swift:35-368), as signature keys
"name(label:label:...)" (or "var name"), plus the `...Async` twins / overloads the protocol declares for the
operations this repository accelerates (HeScheme.swift).  Run in the build container (reads /root/reference).

    python tests/golden/make_noop_scheme_members.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swift_source import static_members  # noqa: E402

REFERENCE = "/root/reference/Sources/HomomorphicEncryption"
noop = static_members(open(os.path.join(REFERENCE, "NoOpScheme.swift")).read(), "NoOpScheme")
protocol = static_members(open(os.path.join(REFERENCE, "HeScheme.swift")).read(), "HeScheme", kind="protocol")
ACCELERATED = ["mulAssign", "relinearize", "modSwitchDown", "modSwitchDownToSingle", "innerProduct", "applyGalois",
               "forwardNtt", "inverseNtt"]
wanted = sorted(m for m in protocol
                if any(m.split("(")[0] in (name, name + "Async") for name in ACCELERATED))
out = {"source": "NoOpScheme.swift:35-368, HeScheme.swift:190-1090 (reference checkout of this build)",
       "noop_scheme_static_members": sorted(noop), "accelerated_protocol_requirements": wanted}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "noop_scheme_members.json")
json.dump(out, open(path, "w"), indent=1)
print(len(noop), "NoOpScheme members,", len(wanted), "accelerated protocol requirements ->", path)
