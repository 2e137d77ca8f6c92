"""testfixtures.TestSchedulingConfig() as data (testfixtures.go:225-249); shared by the scenario drivers."""
import math


def test_scheduling_config():
    pcs = {
        "priority-0": {"priority": 0, "preemptible": True}, "priority-1": {"priority": 1, "preemptible": True},
        "priority-2": {"priority": 2, "preemptible": True}, "priority-2-non-preemptible": {"priority": 2, "preemptible": False},
        "priority-3": {"priority": 3, "preemptible": False},
        "armada-preemptible-away": {"priority": 30000, "preemptible": True, "away": [[29000, "gpu"], [29000, "large"]]},
        "armada-preemptible-away-lower": {"priority": 30000, "preemptible": True, "away": [[28000, "gpu"], [28000, "large"]]},
        "armada-preemptible": {"priority": 30000, "preemptible": True},
    }
    return {
        "priority_classes": pcs,
        "maximum_scheduling_rate": math.inf, "maximum_scheduling_burst": 2**62,
        "maximum_per_queue_scheduling_rate": math.inf, "maximum_per_queue_scheduling_burst": 2**62,
        "indexed_resources": [["cpu", 1000], ["memory", 128 * 2**20], ["nvidia.com/gpu", 1000]],
        "indexed_node_labels": ["largeJobsOnly", "gpu", "cluster", "pool", "nodetype"],
        "indexed_taints": ["largeJobsOnly", "gpu"],
        "well_known_node_types": {"gpu": [["gpu", "true", "NoSchedule"]], "large": [["largeJobsOnly", "true", "NoSchedule"]]},
        "prefer_large_job_ordering": True, "drf_resources": ["cpu", "memory", "nvidia.com/gpu"],
        "protected_fraction_of_fair_share": 0.0, "max_queue_lookback": 0, "maximum_resource_fraction_to_schedule": {},
        "disable_home": False, "disable_away": False, "disable_gang_away": False, "disable_fairshare": False, "disable_urgency": False,
        "disallowed_resources": [],
    }
