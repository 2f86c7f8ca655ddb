"""The SQLite sink against the contract of the reference's
add_message / purge_old_messages (receiver/message_store.c:59-97, 220-262) and
its schema (receiver/generate_db.sql:3-8).  CPU only; the databases are read
back with python's own sqlite3 module, i.e. as the untouched web server would."""
import calendar
import os
import sqlite3
import subprocess
import sys
import time
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
T0 = calendar.timegm((2025, 3, 7, 12, 34, 56))


def rows(path, sql="select bbbb,message,timestamp,age,freq from messages order by id"):
    con = sqlite3.connect(path)
    try:
        return con.execute(sql).fetchall()
    finally:
        con.close()


def test_schema_matches_generate_db_sql(nv, tmp_path):
    db = str(tmp_path / "Navtex.db")
    with nv.Store(db):
        pass
    cols = rows(db, "select name,type,pk from pragma_table_info('messages') order by cid")
    assert cols == [("id", "INTEGER", 1), ("bbbb", "TEXT", 0), ("message", "TEXT", 0), ("timestamp", "TEXT", 0),
                    ("age", "TEXT", 0), ("freq", "INTEGER", 0)]
    assert rows(db, "select id,tag,value from config order by id") == [
        (1, "stations518", "PSTV"), (2, "messages518", "ABCDEFL"), (3, "stations490", "B"), (4, "messages490", "ABCDEFL")]
    # autoincrement ids, as the web server's "messages.id as id" expects
    assert rows(db, "select sql from sqlite_master where name='messages'")[0][0].lower().count("autoincrement") == 1


def test_add_message_semantics(nv, tmp_path):
    db = str(tmp_path / "Navtex.db")
    with nv.Store(db) as st:
        st.set_time(T0)
        assert st.add_message("EA01", "ZCZC EA01\nTEST MESSAGE 123 OK\nNNNN\n", 518) == 0
        assert st.add_message("BB12", "ZCZC BB12\nLOCAL\nNNNN\n", 490) == 0
        st.set_time(T0 + 3600)
        assert st.add_message("EA01", "ZCZC EA01\nREPEAT\nNNNN\n", 518) == 0      # replaces the earlier EA01
        assert st.add_message("", "partial text after an abort\n", 518) == 0      # bbbb may be empty (nav_b_sm.C:49)
        assert st.stats() == (4, 0)
    got = rows(db)
    assert got == [("BB12", "ZCZC BB12\nLOCAL\nNNNN\n", "2025-03-07 12:34", "NEW", 490),
                   ("EA01", "ZCZC EA01\nREPEAT\nNNNN\n", "2025-03-07 13:34", "NEW", 518),
                   ("", "partial text after an abort\n", "2025-03-07 13:34", "NEW", 518)]
    # the reference web server's list query (message_store.c:119) finds what its config admits:
    # station E is not in 'PSTV', B/490 is; the empty id of an aborted message always passes (instr(x,'') = 1)
    listed = rows(db, "select bbbb,freq from messages, config AS CO1, config AS CO2 where "
                      "(freq=518 and CO1.tag='stations518' and instr(CO1.value,substr(bbbb,1,1))>0 and CO2.tag='messages518' and instr(CO2.value,substr(bbbb,2,1)) > 0) OR "
                      "(freq=490 and CO1.tag='stations490' and instr(CO1.value,substr(bbbb,1,1))>0 and CO2.tag='messages490' and instr(CO2.value,substr(bbbb,2,1)) > 0) "
                      "order by timestamp desc")
    assert listed == [("", 518), ("BB12", 490)]


def test_existing_database_is_kept(nv, tmp_path):
    db = str(tmp_path / "Navtex.db")
    con = sqlite3.connect(db)
    con.executescript("""CREATE TABLE IF NOT EXISTS "messages" (id integer primary key autoincrement,bbbb text,message text,timestamp text, age text, freq integer);
                         CREATE TABLE config (id integer primary key autoincrement,tag text,value text);
                         INSERT INTO config VALUES(1,'stations518','ABC');
                         INSERT INTO messages (bbbb,message,timestamp,age,freq) VALUES('PA11','old','2025-03-01 00:00','NEW',518);""")
    con.commit(); con.close()
    with nv.Store(db, create_schema=True) as st:
        st.set_time(T0)
        st.add_message("PA12", "new", 518)
    assert rows(db, "select tag,value from config") == [("stations518", "ABC")]      # user's configuration untouched
    assert [r[0] for r in rows(db)] == ["PA11", "PA12"]


def test_purge_drops_only_old_rows(nv, tmp_path):
    db = str(tmp_path / "Navtex.db")
    with nv.Store(db) as st:
        for k, age_h in enumerate((100, 73, 71, 1)):
            st.set_time(T0 - age_h * 3600)
            st.add_message(f"PA{k:02d}", "x", 518)
        st.set_time(T0)
        assert st.purge() == 2                      # older than the reference's 72 h
        assert [r[0] for r in rows(db)] == ["PA02", "PA03"]
        assert st.purge(1800) == 2                  # explicit limit
        assert st.purge() == 0
    assert rows(db) == []


def test_wall_clock_timestamp_is_utc_minutes(nv, tmp_path):
    db = str(tmp_path / "Navtex.db")
    with nv.Store(db) as st:
        before = time.gmtime()
        st.add_message("PA01", "x", 518)
        after = time.gmtime()
    ts = rows(db)[0][2]
    assert ts in {time.strftime("%Y-%m-%d %H:%M", before), time.strftime("%Y-%m-%d %H:%M", after)}


def test_error_paths(nv, tmp_path):
    with pytest.raises(nv.NvxError):
        nv.Store(str(tmp_path / "no_such_dir" / "x.db"))
    empty = str(tmp_path / "empty.db")
    sqlite3.connect(empty).close()
    with pytest.raises(nv.NvxError):                # without create_schema a database must already have the table
        nv.Store(empty, create_schema=False)
    db = str(tmp_path / "ro.db")
    with nv.Store(db) as st:
        os.chmod(db, 0o444)
        if os.geteuid() != 0:                       # root ignores file modes
            assert st.add_message("PA01", "x", 518) == -2
            assert st.stats() == (0, 1)
    assert nv.lib.nvx_store_add_message(None, b"a", b"b", 1) == -1


def test_weak_add_message_writes_to_NAVTEX_AMD_DB(nv, tmp_path):
    """A program that links the library without the reference's message_store.o gets the
    library's weak add_message; with NAVTEX_AMD_DB set it stores into that database."""
    db = str(tmp_path / "Navtex.db")
    code = ("import ctypes as C, sys; lib = C.CDLL(sys.argv[1]); lib.add_message.argtypes = [C.c_char_p, C.c_char_p, C.c_int];"
            "sys.exit(lib.add_message(b'PA77', b'ZCZC PA77\\nVIA WEAK SINK\\nNNNN\\n', 490))")
    env = dict(os.environ, NAVTEX_AMD_DB=db)
    subprocess.run([sys.executable, "-c", code, str(ROOT / "navtex_amd" / "libnavtex_amd.so")], check=True, env=env)
    assert [(r[0], r[1], r[4]) for r in rows(db)] == [("PA77", "ZCZC PA77\nVIA WEAK SINK\nNNNN\n", 490)]


def test_one_store_shared_by_several_threads(nv, tmp_path):
    """Several handles (threads) may share one store: every message arrives, none twice."""
    import threading
    db = str(tmp_path / "Navtex.db")
    with nv.Store(db) as st:
        st.set_time(T0)
        def worker(k):
            for i in range(25):
                assert st.add_message(f"{chr(65 + k)}A{i:02d}", f"ZCZC {k} {i}\nNNNN\n", 518 if k & 1 else 490) == 0
        threads = [threading.Thread(target=worker, args=(k,)) for k in range(6)]
        for t in threads: t.start()
        for t in threads: t.join()
        assert st.stats() == (150, 0)
    got = rows(db, "select bbbb from messages order by bbbb")
    assert len(got) == 150 and len(set(got)) == 150
