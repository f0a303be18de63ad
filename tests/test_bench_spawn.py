"""`python bench.py --gpus N` typed plainly starts its own N ranks (bench.py::_spawn_ranks): one child per GPU with RANK /
LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR / MASTER_PORT / HSA_ENABLE_IPC_MODE_LEGACY in its environment, rank 0's
stdout forwarded, the worst child code returned.  Checked with a stub child (no GPU, no torch)."""
import argparse
import json
import os
import sys
import textwrap

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def test_spawn_ranks_gives_every_child_its_rank_environment(tmp_path, capsys, monkeypatch):
    import bench

    stub = tmp_path / "child.py"
    stub.write_text(textwrap.dedent("""
        import json, os, sys
        keys = ["RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY"]
        rec = {k: os.environ.get(k) for k in keys}
        rec["argv"] = sys.argv[1:]
        open(os.path.join(sys.argv[1], "rank%s.json" % rec["RANK"]), "w").write(json.dumps(rec))
        print(json.dumps({"from_rank": rec["RANK"]}))
        sys.exit(3 if rec["RANK"] == "5" and len(sys.argv) > 2 and sys.argv[2] == "fail" else 0)
    """))
    monkeypatch.delenv("MASTER_PORT", raising=False)
    args = argparse.Namespace(gpus=8)
    rc = bench._spawn_ranks(args, script=str(stub), argv=[str(tmp_path)])
    out = capsys.readouterr().out.strip().splitlines()
    assert rc == 0 and out == ['{"from_rank": "0"}']  # only rank 0's line reaches the driver
    recs = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(8)]
    assert [r["RANK"] for r in recs] == [str(i) for i in range(8)] and [r["LOCAL_RANK"] for r in recs] == [str(i) for i in range(8)]
    assert {r["WORLD_SIZE"] for r in recs} == {"8"} and {r["LOCAL_WORLD_SIZE"] for r in recs} == {"8"}
    assert {r["MASTER_ADDR"] for r in recs} == {"127.0.0.1"} and {r["HSA_ENABLE_IPC_MODE_LEGACY"] for r in recs} == {"0"}
    ports = {r["MASTER_PORT"] for r in recs}
    assert len(ports) == 1 and 1024 < int(ports.pop()) < 65536  # one rendezvous port, picked free
    # a failing rank fails the run; a MASTER_PORT given by the caller is kept
    monkeypatch.setenv("MASTER_PORT", "29417")
    rc = bench._spawn_ranks(args, script=str(stub), argv=[str(tmp_path), "fail"])
    capsys.readouterr()
    assert rc == 3 and json.load(open(tmp_path / "rank2.json"))["MASTER_PORT"] == "29417"
