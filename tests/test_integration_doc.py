"""INTEGRATION.md shows the ctypes stub a binder would write.  A stub whose struct is shorter
than the C struct hands the library garbage pointers, so the documented field lists are checked
against include/vlnce_hip.h and against the binding the package itself uses."""
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_struct_fields():
    src = open(os.path.join(REPO, "include", "vlnce_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    out = {}
    for body, name in re.findall(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for piece in decl.split(","):
                ident = re.findall(r"[A-Za-z_]\w*", piece)
                fields.append(ident[-1])
        out[name] = fields
    return out


def doc_struct_fields():
    doc = open(os.path.join(REPO, "INTEGRATION.md")).read()
    out = {}
    for m in re.finditer(r"class\s+\w+\(C\.Structure\):\s*#\s*(\w+)\s*\n(.*?)(?=\nclass |\n\n)", doc,
                         flags=re.S):
        body = m.group(2)
        names = re.findall(r"\(\s*\"(\w+)\"\s*,", body)
        tup = re.search(r"for n in \(([^)]*)\)", body, flags=re.S)
        if tup:
            names = re.findall(r"\"(\w+)\"", tup.group(1))
        out[m.group(1)] = names
    return out


def test_documented_stubs_match_the_header():
    hdr, doc = header_struct_fields(), doc_struct_fields()
    assert {"vlnce_conv_desc", "vlnce_prologue", "vlnce_epilogue"} <= set(doc), doc.keys()
    for name, fields in doc.items():
        assert name in hdr, f"INTEGRATION.md documents unknown struct {name}"
        assert fields == hdr[name], (name, fields, hdr[name])


def test_package_binding_matches_the_header():
    from vlnce_amd import _lib

    hdr = header_struct_fields()
    for cls, cname in ((_lib.ConvDesc, "vlnce_conv_desc"), (_lib.Prologue, "vlnce_prologue"),
                       (_lib.Epilogue, "vlnce_epilogue"), (_lib.BnSums, "vlnce_bn_sums")):
        assert [f for f, _ in cls._fields_] == hdr[cname], cname


def test_doc_does_not_claim_memset():
    doc = open(os.path.join(REPO, "INTEGRATION.md")).read()
    assert "`hipMemsetAsync` only" not in doc
