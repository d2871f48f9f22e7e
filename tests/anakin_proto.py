"""The reference's model-file schema for the OFFICIAL protobuf runtime (python google.protobuf), built as descriptors in code - protoc does not
exist in this image. Field numbers / types / labels are those of /root/reference/framework/model_parser/proto/{graph,node,tensor,operator}.proto
(proto3). Test infrastructure: the independent implementation of the wire format that integration/mi355x/framework/anakin_bin_model.h and
anakin_amd/anakin_bin.py are checked against (tests/test_anakin_bin.py)."""
from google.protobuf import descriptor_pb2 as D
from google.protobuf import descriptor_pool, message_factory

F = D.FieldDescriptorProto
OPT, REP = F.LABEL_OPTIONAL, F.LABEL_REPEATED


def _field(msg, name, number, ftype, label=OPT, type_name=None, oneof=None):
    f = msg.field.add()
    f.name, f.number, f.type, f.label = name, number, ftype, label
    if type_name:
        f.type_name = type_name
    if oneof is not None:
        f.oneof_index = oneof
    return f


def _map(msg, name, number, value_type_name):
    """map<string, V> name = number;  ->  repeated NameEntry { string key = 1; V value = 2; } with map_entry"""
    entry = msg.nested_type.add()
    entry.name = "".join(p.capitalize() for p in name.split("_")) + "Entry"
    entry.options.map_entry = True
    _field(entry, "key", 1, F.TYPE_STRING)
    _field(entry, "value", 2, F.TYPE_MESSAGE, type_name=value_type_name)
    _field(msg, name, number, F.TYPE_MESSAGE, REP, type_name=".%s.%s" % (msg.name, entry.name))


def build():
    pool = descriptor_pool.DescriptorPool()

    fd = D.FileDescriptorProto()                       # tensor.proto
    fd.name, fd.syntax = "tensor.proto", "proto3"
    ts = fd.message_type.add(); ts.name = "TensorShape"
    dim = ts.nested_type.add(); dim.name = "Dim"
    _field(dim, "value", 1, F.TYPE_INT32, REP)
    _field(dim, "size", 2, F.TYPE_INT64)
    _field(ts, "dim", 3, F.TYPE_MESSAGE, type_name=".TensorShape.Dim")
    e = fd.enum_type.add(); e.name = "DateTypeProto"
    for n, v in (("STR", 0), ("INT8", 2), ("INT32", 4), ("FLOAT16", 8), ("FLOAT", 13), ("DOUBLE", 14), ("BOOLEN", 20), ("CACHE_LIST", 30), ("TENSOR", 31)):
        x = e.value.add(); x.name, x.number = n, v
    cd = fd.message_type.add(); cd.name = "CacheDate"
    _field(cd, "s", 1, F.TYPE_BYTES, REP)
    _field(cd, "i", 2, F.TYPE_INT32, REP)
    _field(cd, "f", 3, F.TYPE_FLOAT, REP)
    _field(cd, "b", 4, F.TYPE_BOOL, REP)
    _field(cd, "l", 5, F.TYPE_MESSAGE, REP, type_name=".CacheDate")
    _field(cd, "c", 8, F.TYPE_BYTES)
    _field(cd, "type", 6, F.TYPE_ENUM, type_name=".DateTypeProto")
    _field(cd, "size", 7, F.TYPE_INT64)
    tp = fd.message_type.add(); tp.name = "TensorProto"
    _field(tp, "name", 1, F.TYPE_BYTES)
    _field(tp, "shared", 2, F.TYPE_BOOL)
    _field(tp, "share_from", 3, F.TYPE_BYTES)
    _field(tp, "shape", 8, F.TYPE_MESSAGE, type_name=".TensorShape")
    _field(tp, "valid_shape", 9, F.TYPE_MESSAGE, type_name=".TensorShape")
    _field(tp, "data", 10, F.TYPE_MESSAGE, type_name=".CacheDate")
    _field(tp, "scale", 11, F.TYPE_MESSAGE, type_name=".CacheDate")
    pool.Add(fd)

    fd = D.FileDescriptorProto()                       # operator.proto
    fd.name, fd.syntax = "operator.proto", "proto3"
    op = fd.message_type.add(); op.name = "OpProto"
    _field(op, "name", 1, F.TYPE_STRING)
    _field(op, "is_commutative", 2, F.TYPE_BOOL)
    _field(op, "in_num", 3, F.TYPE_INT32)
    _field(op, "out_num", 4, F.TYPE_INT32)
    _field(op, "description", 5, F.TYPE_STRING)
    pool.Add(fd)

    fd = D.FileDescriptorProto()                       # node.proto
    fd.name, fd.syntax = "node.proto", "proto3"
    fd.dependency.extend(["operator.proto", "tensor.proto"])
    vt = fd.message_type.add(); vt.name = "valueType"
    vt.oneof_decl.add().name = "data"
    _field(vt, "s", 1, F.TYPE_BYTES, oneof=0)
    _field(vt, "i", 2, F.TYPE_INT32, oneof=0)
    _field(vt, "f", 3, F.TYPE_FLOAT, oneof=0)
    _field(vt, "b", 4, F.TYPE_BOOL, oneof=0)
    _field(vt, "cache_list", 8, F.TYPE_MESSAGE, type_name=".CacheDate", oneof=0)
    _field(vt, "tensor", 10, F.TYPE_MESSAGE, type_name=".TensorProto", oneof=0)
    _field(vt, "type", 14, F.TYPE_ENUM, type_name=".DateTypeProto")
    nd = fd.message_type.add(); nd.name = "NodeProto"
    _field(nd, "name", 1, F.TYPE_STRING)
    _field(nd, "ins", 2, F.TYPE_STRING, REP)
    _field(nd, "outs", 3, F.TYPE_STRING, REP)
    _map(nd, "attr", 10, ".valueType")
    _field(nd, "lane", 11, F.TYPE_INT32)
    _field(nd, "need_wait", 12, F.TYPE_BOOL)
    _field(nd, "Op", 15, F.TYPE_MESSAGE, type_name=".OpProto")
    _field(nd, "bit_type", 16, F.TYPE_ENUM, type_name=".DateTypeProto")
    pool.Add(fd)

    fd = D.FileDescriptorProto()                       # graph.proto
    fd.name, fd.syntax = "graph.proto", "proto3"
    fd.dependency.extend(["node.proto", "tensor.proto"])
    ver = fd.message_type.add(); ver.name = "Version"
    _field(ver, "major", 1, F.TYPE_INT32); _field(ver, "minor", 2, F.TYPE_INT32); _field(ver, "patch", 3, F.TYPE_INT32)
    _field(ver, "version", 4, F.TYPE_INT64)
    info = fd.message_type.add(); info.name = "Info"
    _field(info, "temp_mem_used", 1, F.TYPE_INT32); _field(info, "original_temp_mem_used", 2, F.TYPE_INT32)
    _field(info, "system_mem_used", 3, F.TYPE_INT32); _field(info, "model_mem_used", 4, F.TYPE_INT32)
    _field(info, "is_optimized", 10, F.TYPE_BOOL)
    lp = fd.enum_type.add(); lp.name = "LayoutProto"
    for v, n in enumerate(("Invalid", "LP_W", "LP_HW", "LP_WH", "LP_NC", "LP_NH", "LP_NW", "LP_NHW", "LP_NCHW", "LP_NHWC", "LP_NCHW_C4", "LP_NCHW_C8",
                           "LP_NCHW_C16", "LP_OIHW16I16O", "LP_GOIHW16I16O", "LP_NCHW_C8R", "LP_NCHW_C16R")):
        x = lp.value.add(); x.name, x.number = n, v
    tg = fd.message_type.add(); tg.name = "TargetProto"
    _field(tg, "node", 1, F.TYPE_STRING); _field(tg, "scale", 2, F.TYPE_FLOAT, REP); _field(tg, "layout", 3, F.TYPE_ENUM, type_name=".LayoutProto")
    ls = fd.message_type.add(); ls.name = "List"
    _field(ls, "val", 1, F.TYPE_STRING, REP); _field(ls, "target", 2, F.TYPE_MESSAGE, REP, type_name=".TargetProto")
    gp = fd.message_type.add(); gp.name = "GraphProto"
    _field(gp, "name", 1, F.TYPE_STRING)
    _field(gp, "nodes", 2, F.TYPE_MESSAGE, REP, type_name=".NodeProto")
    _map(gp, "edges_in", 3, ".List")
    _map(gp, "edges_out", 4, ".List")
    _map(gp, "edges_info", 5, ".TensorProto")
    _field(gp, "ins", 6, F.TYPE_STRING, REP)
    _field(gp, "outs", 7, F.TYPE_STRING, REP)
    _field(gp, "version", 10, F.TYPE_MESSAGE, type_name=".Version")
    _field(gp, "summary", 11, F.TYPE_MESSAGE, type_name=".Info")
    pool.Add(fd)

    names = ("GraphProto", "NodeProto", "valueType", "TensorProto", "CacheDate", "TensorShape", "OpProto", "List", "TargetProto", "Version", "Info")
    return {n: message_factory.GetMessageClass(pool.FindMessageTypeByName(n)) for n in names}


STR, INT8, INT32, FLOAT16, FLOAT, DOUBLE, BOOLEN, CACHE_LIST, TENSOR = 0, 2, 4, 8, 13, 14, 20, 30, 31
