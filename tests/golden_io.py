import json
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLDEN, f"{name}_cases.json")) as f:
        return json.load(f)


def ids(cases):
    return [c["name"] for c in cases]
