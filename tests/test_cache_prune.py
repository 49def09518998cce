"""The kernel cache stays small enough to travel: __graft_entry__.prune_kernel_cache removes code objects that the current build
did not use (libexahip re-stamps a cache file it loads), oldest first, down to a size limit — and nothing else."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_prune_removes_only_unused_code_objects_oldest_first(tmp_path):
    import __graft_entry__ as ge
    now = time.time()
    spec = [("old1.hsaco", 400_000, now - 300), ("old2.hsaco", 400_000, now - 200), ("old3.hsaco", 400_000, now - 100),
            ("used.hsaco", 400_000, now + 1), ("old.note", 10, now - 300), ("old.tune", 10, now - 300)]
    for name, size, mtime in spec:
        p = tmp_path / name
        p.write_bytes(b"x" * size)
        os.utime(p, (mtime, mtime))
    total = ge.prune_kernel_cache(str(tmp_path), now, limit_mb=1.0)          # 1.6 MB -> two oldest go
    left = sorted(os.listdir(tmp_path))
    assert left == ["old.note", "old.tune", "old3.hsaco", "used.hsaco"], left
    assert total <= 1.0e6
    # below the limit: nothing happens; a limit nothing unused can satisfy: the used ones stay
    assert ge.prune_kernel_cache(str(tmp_path), now, limit_mb=1.0) == total
    ge.prune_kernel_cache(str(tmp_path), now, limit_mb=0.0)
    assert sorted(os.listdir(tmp_path)) == ["old.note", "old.tune", "used.hsaco"]


def test_loading_a_cached_module_stamps_it(tmp_path, monkeypatch):
    """A disk hit re-stamps the code object (that is what the pruner goes by)."""
    sys.path.insert(0, os.path.join(ROOT, "examodels.jl_amd"))
    from exahip import ExaModel, models
    monkeypatch.setenv("EXAHIP_CACHE_DIR", str(tmp_path))
    path = ExaModel(models.luksan_vlcek_model(50), device=False).compile()
    assert os.path.dirname(path) == str(tmp_path)
    os.utime(path, (1_000_000_000, 1_000_000_000))
    assert ExaModel(models.luksan_vlcek_model(60), device=False).compile() == path       # same module for every N: a disk hit
    assert os.stat(path).st_mtime > 1_700_000_000
