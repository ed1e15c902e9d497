"""knossos.model -- the Model protocol and the stock models (recalled,
SURVEY.md section 8a): register, cas-register, mutex, multi-register, set; plus the
bank model Knossos does not ship, specified from the reference's own
ledger->bank mapping (/root/reference/src/tigerbeetle/tests/ledger.clj:89-114).

`step(op)` returns the next model or an `Inconsistent`; models are immutable
and hashable so knossos.model.memo can enumerate them.  These classes are the
SPECIFICATION the device code is written to and the input of the memo-table
path; the search itself never runs in Python.
"""
from __future__ import annotations


class Inconsistent:
    def __init__(self, msg):
        self.msg = msg

    def step(self, op):
        return self

    def __eq__(self, o):
        return isinstance(o, Inconsistent) and o.msg == self.msg

    def __hash__(self):
        return hash(("inconsistent", self.msg))

    def __repr__(self):
        return f"Inconsistent({self.msg!r})"


def inconsistent(msg):
    return Inconsistent(msg)


def inconsistent_p(m):
    return isinstance(m, Inconsistent)


class Model:
    def step(self, op):  # pragma: no cover
        raise NotImplementedError

    def _key(self):  # pragma: no cover
        raise NotImplementedError

    def __eq__(self, o):
        return type(o) is type(self) and o._key() == self._key()

    def __hash__(self):
        return hash((type(self).__name__, self._key()))


class Register(Model):
    def __init__(self, value=None):
        self.value = value

    def step(self, op):
        f, v = op["f"], op.get("value")
        if f == "write":
            return Register(v)
        if f == "read":
            if v is None or v == self.value:
                return self
            return inconsistent(f"can't read {v} from register {self.value}")
        return inconsistent(f"unknown op {f}")

    def _key(self):
        return self.value

    def __repr__(self):
        return f"Register({self.value!r})"


class CASRegister(Model):
    def __init__(self, value=None):
        self.value = value

    def step(self, op):
        f, v = op["f"], op.get("value")
        if f == "write":
            return CASRegister(v)
        if f == "cas":
            cur, new = v
            if cur == self.value:
                return CASRegister(new)
            return inconsistent(f"can't CAS {self.value} from {cur} to {new}")
        if f == "read":
            if v is None or v == self.value:
                return self
            return inconsistent(f"can't read {v} from register {self.value}")
        return inconsistent(f"unknown op {f}")

    def _key(self):
        return self.value

    def __repr__(self):
        return f"CASRegister({self.value!r})"


class Mutex(Model):
    def __init__(self, locked=False):
        self.locked = locked

    def step(self, op):
        f = op["f"]
        if f == "acquire":
            return inconsistent("already held") if self.locked else Mutex(True)
        if f == "release":
            return Mutex(False) if self.locked else inconsistent("not held")
        return inconsistent(f"unknown op {f}")

    def _key(self):
        return self.locked

    def __repr__(self):
        return f"Mutex({self.locked})"


class MultiRegister(Model):
    """:f :txn, value = [[f k v] ...] with f in r / w; the whole txn is atomic."""

    def __init__(self, values=()):
        self.values = tuple(sorted(dict(values).items())) if not isinstance(values, tuple) else values

    def step(self, op):
        if op["f"] != "txn":
            return inconsistent(f"unknown op {op['f']}")
        st = dict(self.values)
        for f, k, v in op["value"]:
            if f in ("r", "read"):
                if v is not None and st.get(k) != v:
                    return inconsistent(f"can't read {v} from key {k} = {st.get(k)}")
            elif f in ("w", "write"):
                st[k] = v
            else:
                return inconsistent(f"unknown micro-op {f}")
        return MultiRegister(tuple(sorted(st.items())))

    def _key(self):
        return self.values

    def __repr__(self):
        return f"MultiRegister({dict(self.values)!r})"


class SetModel(Model):
    """knossos.model/set: :add v -> conj; :read s ok iff s equals the state exactly."""

    def __init__(self, s=frozenset()):
        self.s = frozenset(s)

    def step(self, op):
        f, v = op["f"], op.get("value")
        if f == "add":
            return SetModel(self.s | {v})
        if f == "read":
            if v is None or frozenset(v) == self.s:
                return self
            return inconsistent(f"can't read {sorted(v)} from {sorted(self.s)}")
        return inconsistent(f"unknown op {f}")

    def _key(self):
        return self.s

    def __repr__(self):
        return f"SetModel({sorted(self.s)!r})"


class Bank(Model):
    """NEW (Knossos has none): accounts -> balance, initial 0 (the reference's
    default :total-amount 0, accounts 1..8: tests/ledger.clj:355-357);
    :transfer {:debit-acct :credit-acct :amount} moves amount (balance =
    credits - debits, tests/ledger.clj:102-103); :read ok iff the map read
    equals the state; with negative_balances False a transfer that would
    overdraw is inconsistent (core.clj:217-219)."""

    def __init__(self, balances, negative_balances=True):
        self.balances = tuple(sorted(dict(balances).items())) if not isinstance(balances, tuple) else balances
        self.neg = negative_balances

    def step(self, op):
        f, v = op["f"], op.get("value")
        st = dict(self.balances)
        if f == "transfer":
            d, c, amt = v["debit-acct"], v["credit-acct"], v["amount"]
            if d not in st or c not in st:
                return inconsistent(f"unknown account in {v}")
            st[d] -= amt
            st[c] += amt
            if not self.neg and st[d] < 0:
                return inconsistent(f"account {d} would go negative")
            return Bank(tuple(sorted(st.items())), self.neg)
        if f == "read":
            if v is None or dict(v) == st:
                return self
            return inconsistent(f"can't read {dict(v)} from {st}")
        return inconsistent(f"unknown op {f}")

    def _key(self):
        return (self.balances, self.neg)

    def __repr__(self):
        return f"Bank({dict(self.balances)!r})"


def register(value=None):
    return Register(value)


def cas_register(value=None):
    return CASRegister(value)


def mutex():
    return Mutex(False)


def multi_register(values=None):
    return MultiRegister(tuple(sorted((values or {}).items())))


def set():  # noqa: A001 - mirrors knossos.model/set
    return SetModel()


def bank(accounts=range(1, 9), negative_balances=True):
    return Bank(tuple((a, 0) for a in accounts), negative_balances)
