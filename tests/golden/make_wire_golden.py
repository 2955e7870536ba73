#!/usr/bin/env python
"""Golden byte strings for the in-tree protobuf messages of the scoring path (microscopes/io/schema.proto:3-46),
serialised by Google's own protobuf runtime -- an implementation independent of include/microscopes_amd/wire.hpp and
common_amd/wire.py, which the tests hold against these bytes.

The message definitions below are the DATA of schema.proto (names, field numbers, labels, types) entered as a
FileDescriptorProto; no protoc is needed.  Contents: fixed values, plus the scenario of the reference's own
test/cxx/test_group_manager.cpp:22-66 for GroupManager.  Writes tests/golden/wire.json.
"""
import json
import os

from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

F = descriptor_pb2.FieldDescriptorProto
REQ, REP = F.LABEL_REQUIRED, F.LABEL_REPEATED


def build_pool():
    fdp = descriptor_pb2.FileDescriptorProto()
    fdp.name, fdp.package, fdp.syntax = "microscopes/io/schema.proto", "microscopes.io", "proto2"

    def msg(parent, name, fields):
        m = (parent.message_type if parent is fdp else parent.nested_type).add()
        m.name = name
        for fname, number, label, ftype, *tn in fields:
            f = m.field.add()
            f.name, f.number, f.label, f.type = fname, number, label, ftype
            if tn:
                f.type_name = tn[0]
        return m

    msg(fdp, "CRP", [("alpha", 1, REQ, F.TYPE_FLOAT)])                                         # schema.proto:3-5
    bbnc = msg(fdp, "BetaBernoulliNonConj", [])                                                # :7-18
    msg(bbnc, "Shared", [("alpha", 1, REQ, F.TYPE_FLOAT), ("beta", 2, REQ, F.TYPE_FLOAT)])
    msg(bbnc, "Group", [("p", 1, REQ, F.TYPE_FLOAT), ("heads", 2, REQ, F.TYPE_UINT32), ("tails", 3, REQ, F.TYPE_UINT32)])
    dm = msg(fdp, "DirichletMultinomial", [])                                                  # :20-29
    msg(dm, "Shared", [("alphas", 1, REP, F.TYPE_FLOAT)])
    msg(dm, "Group", [("counts", 1, REP, F.TYPE_UINT32), ("ratio", 2, REQ, F.TYPE_FLOAT)])
    msg(fdp, "GroupData", [("id", 1, REQ, F.TYPE_UINT32), ("data", 2, REQ, F.TYPE_BYTES)])      # :31-34
    msg(fdp, "GroupManager", [("alpha", 1, REQ, F.TYPE_FLOAT), ("assignments", 2, REP, F.TYPE_INT32),   # :36-40
                              ("groups", 3, REP, F.TYPE_MESSAGE, ".microscopes.io.GroupData")])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fdp)
    return pool


def main():
    pool = build_pool()

    def cls(name):
        return message_factory.GetMessageClass(pool.FindMessageTypeByName("microscopes.io." + name))

    out = []

    def emit(name, fields, m):
        out.append({"message": name, "fields": fields, "hex": m.SerializeToString().hex()})

    for alpha in (2.0, 0.5, 1.5):
        emit("CRP", {"alpha": alpha}, cls("CRP")(alpha=alpha))
    for a, b in ((1.0, 1.0), (1.5, 0.25), (2.0, 7.5)):
        emit("BetaBernoulliNonConj.Shared", {"alpha": a, "beta": b}, cls("BetaBernoulliNonConj.Shared")(alpha=a, beta=b))
    for p, h, t in ((0.25, 3, 300), (0.5, 0, 0), (0.8125, 70000, 1)):
        emit("BetaBernoulliNonConj.Group", {"p": p, "heads": h, "tails": t},
             cls("BetaBernoulliNonConj.Group")(p=p, heads=h, tails=t))
    for alphas in ([1.0, 2.0], [0.5, 1.5, 2.5, 0.125], [1.0] * 5):
        emit("DirichletMultinomial.Shared", {"alphas": alphas}, cls("DirichletMultinomial.Shared")(alphas=alphas))
    for counts, ratio in (([1, 128], 1.0), ([0, 0, 0], 0.0), ([5, 300, 70000, 2], 12.75)):
        emit("DirichletMultinomial.Group", {"counts": counts, "ratio": ratio},
             cls("DirichletMultinomial.Group")(counts=counts, ratio=ratio))
    # test/cxx/test_group_manager.cpp:22-66: 10 entities, alpha 2, groups 0..6 created, 3 deleted, the assignment
    # vector below; a group's data is the number of add_value calls it saw, serialised with to_string
    assignments = [-1, 2, 1, 0, 6, 1, 2, -1, -1, 5]
    gm = cls("GroupManager")(alpha=2.0, assignments=assignments)
    data = {}
    for gid in (0, 1, 2, 4, 5, 6):
        data[gid] = str(sum(1 for a in assignments if a == gid))
        g = gm.groups.add()
        g.id, g.data = gid, data[gid].encode()
    emit("GroupManager", {"alpha": 2.0, "assignments": assignments, "groups": {str(k): v for k, v in data.items()},
                          "scenario": "test/cxx/test_group_manager.cpp:22-66"}, gm)
    emit("GroupData", {"id": 7, "data_hex": "0801"}, cls("GroupData")(id=7, data=bytes.fromhex("0801")))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wire.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
        fh.write("\n")
    print("wrote", path, len(out), "vectors")


if __name__ == "__main__":
    main()
