"""paragraph_amd.graph_templates against the graphs the reference's templates produced for the five events of
share/test-data/multiparagraph/candidates.json, as recorded in expected.json (tests/golden/sites/multiparagraph)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_templates_reproduce_the_recorded_graphs():
    from paragraph_amd import graph_templates
    expected = json.load(open(os.path.join(ROOT, "tests", "golden", "sites", "multiparagraph", "expected.json")))
    kinds = []
    for record in expected:
        event = {k: record[k] for k in ("chrom", "start", "end", "ins", "flank") if k in record}
        kind, graph = graph_templates.make_graph(event)
        kinds.append(kind)
        assert kind == record["type"], record["desc"]
        for key in ("nodes", "edges", "paths", "sequencenames", "target_regions"):
            assert graph[key] == record["graph"][key], (record["desc"], key)
        assert set(graph) == {"nodes", "edges", "paths", "sequencenames", "target_regions"}
    assert kinds == ["del", "swap", "longdel", "longswap", "ins"]


def test_template_input_checks():
    import pytest
    from paragraph_amd import graph_templates
    with pytest.raises(ValueError):
        graph_templates.make_graph({"chrom": "c", "start": 10, "end": 9})
    kind, g = graph_templates.make_graph({"chrom": "c", "start": 500, "end": 1200})  # default flank 150: long
    assert kind == "longdel" and g["target_regions"] == ["c:349-651", "c:1049-1351"]
