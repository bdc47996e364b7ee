"""root_digger_amd.cli is a launcher of bin/rd_amd (the reference's `rd` command line is ONE native
program, csrc/tools/rd_main.cpp = /root/reference/src/main.cpp:411-680): arguments pass through,
`--workers` is the reference's `--threads`, `--gpus N` starts N ranks with the rank environment a
launcher would set, rank 0's stdout is relayed, the first non-zero exit code is returned.  Checked
here on the CPU with a stand-in program in rd_amd's place."""
import json
import os
import stat

from root_digger_amd import cli


def _stand_in(tmp_path, exit_code_of_rank_1=0):
    exe = tmp_path / "fake_rd_amd"
    exe.write_text("""#!/bin/bash
echo "{\\"rank\\": \\"${RANK:-none}\\", \\"world\\": \\"${WORLD_SIZE:-none}\\", \\"local\\": \\"${LOCAL_RANK:-none}\\", \\"addr\\": \\"${MASTER_ADDR:-none}\\", \\"port\\": \\"${MASTER_PORT:-none}\\", \\"args\\": \\"$*\\"}" >> %s/ranks.jsonl
echo "rank ${RANK:-0} says: $*"
if [ "${RANK:-0}" = "1" ]; then exit %d; fi
exit 0
""" % (tmp_path, exit_code_of_rank_1))
    exe.chmod(exe.stat().st_mode | stat.S_IEXEC)
    return str(exe)


def test_arguments_pass_through_and_workers_is_threads(tmp_path, monkeypatch, capsys):
    monkeypatch.setattr(cli, "RD_AMD", _stand_in(tmp_path))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert cli.main(["--msa", "a.fasta", "--tree", "t.nwk", "--exhaustive", "--workers", "3", "--early-stop"]) == 0
    out = capsys.readouterr().out
    assert out == "rank 0 says: --msa a.fasta --tree t.nwk --exhaustive --threads 3 --early-stop\n"   # relayed through sys.stdout
    rec = [json.loads(l) for l in open(tmp_path / "ranks.jsonl")]
    assert len(rec) == 1 and rec[0]["world"] == "none"          # one process, no rank environment invented


def test_gpus_starts_the_ranks_with_a_launchers_environment(tmp_path, monkeypatch, capsys):
    monkeypatch.setattr(cli, "RD_AMD", _stand_in(tmp_path))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert cli.main(["--msa", "a.fasta", "--gpus", "3", "--site-shards", "3"]) == 0
    assert capsys.readouterr().out == "rank 0 says: --msa a.fasta --site-shards 3\n"              # rank 0 only
    rec = sorted((json.loads(l) for l in open(tmp_path / "ranks.jsonl")), key=lambda r: r["rank"])
    assert [r["rank"] for r in rec] == ["0", "1", "2"] and [r["local"] for r in rec] == ["0", "1", "2"]
    assert {r["world"] for r in rec} == {"3"} and {r["addr"] for r in rec} == {"127.0.0.1"}
    assert len({r["port"] for r in rec}) == 1 and rec[0]["port"].isdigit()
    assert all(r["args"] == "--msa a.fasta --site-shards 3" for r in rec)                        # --gpus is the launcher's own


def test_under_a_launcher_the_environment_is_left_alone_and_failures_come_back(tmp_path, monkeypatch, capsys):
    monkeypatch.setattr(cli, "RD_AMD", _stand_in(tmp_path, exit_code_of_rank_1=7))
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setenv("LOCAL_RANK", "1")
    assert cli.main(["--msa", "a.fasta", "--gpus", "2"]) == 7          # this process IS rank 1 of a launcher's two
    rec = [json.loads(l) for l in open(tmp_path / "ranks.jsonl")]
    assert len(rec) == 1 and rec[0]["rank"] == "1" and rec[0]["world"] == "2"
    # without a launcher: rank 1 of the two it starts fails -> its code is the launcher's
    os.remove(tmp_path / "ranks.jsonl")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k)
    assert cli.main(["--msa", "a.fasta", "--gpus", "2"]) == 7
