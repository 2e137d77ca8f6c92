#!/usr/bin/env python3
"""Turns the reference simulator's own input files into the JSON fixture tests/golden/simulator_basic_input.json.

BASELINE configs[0] ("cmd/simulator: 100 nodes, 1 queue, 1k single-pod jobs") is the cluster testdata/clusters/cpu_1_1_100.yaml, the
workload shape of testdata/workloads/basicWorkload.yaml with `number: 1000` and testdata/configs/basicSchedulingConfig.yaml
(maximumResourceFractionToSchedule 0.025 on cpu and memory).  /root/reference does not exist on the GPU box, so the three documents are
parsed HERE and stored verbatim (as data, with their source paths) in one JSON file; armada_amd.simulator_input reads either form.

    python tests/golden/make_simulator_fixture.py
"""
import json
import os

import yaml

REF = "/root/reference/internal/scheduler/simulator/testdata"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    docs = {}
    for key, rel in (("cluster", "clusters/cpu_1_1_100.yaml"), ("workload", "workloads/basicWorkload.yaml"), ("config", "configs/basicSchedulingConfig.yaml")):
        with open(os.path.join(REF, rel)) as f:
            docs[key] = {"source": "internal/scheduler/simulator/testdata/" + rel, "doc": yaml.safe_load(f)}
    # BASELINE configs[0]: 1k single-pod jobs (basicWorkload.yaml ships number: 10)
    docs["workload"]["doc"]["queues"][0]["jobTemplates"][0]["number"] = 1000
    docs["workload"]["override"] = "queues[0].jobTemplates[0].number = 1000 (BASELINE configs[0]; the file ships 10)"
    out = os.path.join(HERE, "simulator_basic_input.json")
    with open(out, "w") as f:
        json.dump(docs, f, indent=1, sort_keys=True)
    print("wrote", out)


if __name__ == "__main__":
    main()
