"""EDN history reader / writer: the op shapes the reference documents
(README.md:41-50,67-74,99-106; set_full.clj:29-31,42-45,107-110; workloads/ledger.clj)."""
import os

import pytest

from jepsen_tigerbeetle_amd.jepsen import edn, independent
from jepsen_tigerbeetle_amd.knossos import history as H

SAMPLE = r'''
{:type :invoke, :f :add, :value [1 9], :process 0, :time 3291485317, :index 0}
{:type :ok, :f :add, :value [1 9], :process 0, :time 3341242517, :index 1, :node "n1", :client [0 0]}
{:type :invoke, :f :read, :value [1 nil], :process 1, :time 3400000000, :index 2}
; a nemesis line, README.md:105
{:process :nemesis, :type :info, :f :start-partition, :value [:isolated {"n1" #{"n2" "n3"}, "n2" #{"n1"}, "n3" #{"n1"}}], :index 3}
{:type :ok, :f :read, :value [1 #{9}], :process 1, :time 3500000000, :index 4, :final? true}
{:type :info, :f :add, :value [2 10], :process 2, :error :timeout, :index 5}
#jepsen.history.Op{:type :invoke, :f :txn, :value [[:t 7 {:debit-acct 2, :credit-acct 1, :amount 3}]], :process 3, :index 6}
{:type :fail, :f :cas, :value [1 2], :process 4, :index 7, :error "can't \"cas\""}
'''


def test_reads_jepsen_op_shapes():
    ops = edn.read_history(SAMPLE.strip().splitlines())
    assert len(ops) == 8
    assert ops[0] == {"type": "invoke", "f": "add", "value": [1, 9], "process": 0, "time": 3291485317, "index": 0}
    assert ops[1]["node"] == "n1" and ops[1]["client"] == [0, 0]
    assert ops[3]["process"] == "nemesis" and ops[3]["value"][0] == "isolated"
    assert ops[3]["value"][1]["n1"] == frozenset({"n2", "n3"})
    assert ops[4]["value"] == [1, frozenset({9})] and ops[4]["final?"] is True
    assert ops[5]["error"] == "timeout"
    assert ops[6]["value"] == [["t", 7, {"debit-acct": 2, "credit-acct": 1, "amount": 3}]]
    assert ops[7]["error"] == 'can\'t "cas"'
    assert [o["process"] for o in ops if H.client_op(o)] == [0, 0, 1, 1, 2, 3, 4]
    keyed = edn.read_history(SAMPLE.strip().splitlines(), tuple_keys=True)
    assert isinstance(keyed[0]["value"], independent.Tuple) and keyed[4]["value"].value == frozenset({9})
    assert independent.history_keys(keyed)[:2] == [1, 2]


def test_round_trip(tmp_path):
    hist = [{"type": "invoke", "f": "cas", "value": [1, 2], "process": 0, "index": 0},
            {"type": "ok", "f": "cas", "value": [1, 2], "process": 0, "index": 1},
            {"type": "invoke", "f": "read", "value": None, "process": 1, "index": 2},
            {"type": "info", "f": "read", "value": None, "process": 1, "index": 3, "error": "timeout"},
            {"type": "info", "f": "kill", "value": ["n1", "n2"], "process": "nemesis", "index": 4}]
    p = os.path.join(tmp_path, "history.edn")
    edn.write_history(p, hist)
    assert open(p).readline().startswith("{:type :invoke, :f :cas, :value [1 2], :process 0")
    back = edn.read_history(p)
    assert back == hist


def test_errors():
    with pytest.raises(edn.EDNError):
        edn.loads("{:a 1")
    with pytest.raises(edn.EDNError):
        edn.loads("{:a}")
    assert edn.loads("[1 2.5 -3 1N \\a true nil sym #_ignored :k]") == [1, 2.5, -3, 1, "a", True, None, "sym", "k"]
