"""Option structs and fluent builders with the reference's names.

Mirrors /root/reference/bigseqkit/*.go: ``SeqKitConfig`` (helper.go:25-138) and one
``SeqKit<Cmd>Options`` per hot-path command.  A builder method has the Go name
(``.All(True)``, ``.Tabular(True)``, ``.Config(cfg)``); ``to_json()`` produces the
text of ``bigseqkit.OptionsToString`` (helper.go:47-55): unset pointer fields are
``null`` and are filled by ``setDefaults()`` inside the library (bsk_create).
"""
import json

_CONFIG_FIELDS = ["SeqType", "ChunkSize", "BufferSize", "LineWidth", "IDRegexp", "IDNCBI", "Quiet",
                  "AlphabetGuessSeqLength", "ValidateSeqLength"]

_FIELDS = {
    "Stats": ["Tabular", "GapLetters", "All", "SkipErr", "FqEncoding", "Basename"],
    "SeqTransform": ["Reverse", "Complement", "Name", "Seq", "Qual", "OnlyId", "RemoveGaps", "GapLetters",
                     "LowerCase", "UpperCase", "Dna2rna", "Rna2dna", "ValidateSeq", "ValidateSeqLength", "MaxLen",
                     "MinLen", "QualAsciiBase", "MinQual", "MaxQual"],
    "Grep": ["Pattern", "PatternFile", "UseRegexp", "DeleteMatched", "InvertMatch", "ByName", "BySeq",
             "OnlyPositiveStrand", "MaxMismatch", "IgnoreCase", "Degenerate", "Region", "Circular", "Count"],
    "Locate": ["Pattern", "PatternFile", "Degenerate", "UseRegexp", "UseFmi", "IgnoreCase", "OnlyPositiveStrand",
               "ValidateSeqLength", "NonGreedy", "Gtf", "Bed", "MaxMismatch", "HideMatched", "Circular"],
    "SubseqTransform": ["Chr", "Region", "Gtf", "Feature", "UpStream", "DownStream", "OnlyFlank", "Bed", "GtfTag"],
    "Translate": ["TranslTable", "Frame", "Trim", "Clean", "AllowUnknownCodon", "InitCodonAsM", "ListTranslTable",
                  "ListTranslTableWithAmbCodons", "AppendFrame"],
    "RmDup": ["ByName", "BySeq", "IgnoreCase", "DupSeqsFile", "DupNumFile", "OnlyPositiveStrand"],
    "Fq2Fa": [],              # bigseqkit/fq2fa.go:11-13
    "Range": ["Range"],       # bigseqkit/range.go:14-17
    "Head": ["N"],            # bigseqkit/head.go:12-15
    "Duplicate": ["Times"],   # bigseqkit/duplicate.go:9-12
    "Rename": ["ByName"],     # bigseqkit/rename.go:12-15
    "Pair": ["SaveUnpaired"], # bigseqkit/pair.go:12-15
    "Concat": ["Full", "Separator"],   # bigseqkit/concat.go:12-16
    "Common": ["ByName", "BySeq", "IgnoreCase", "OnlyPositiveStrand"],   # bigseqkit/common.go:13-19
    "Faidx": ["UseRegexp", "IgnoreCase", "FullHead", "RegionFile", "Regions"],   # bigseqkit/faidx.go:11-18
    "Sort": ["InNaturalOrder", "BySeq", "ByName", "ByLength", "ByBases", "GapLetters", "Reverse", "IgnoreCase",
             "SeqPrefixLength"],   # bigseqkit/sort.go:13-24
}


class _Builder:
    _fields = ()

    def __init__(self):
        object.__setattr__(self, "_v", {})

    def __getattr__(self, name):
        if name in self._fields:
            def setter(value, _n=name):
                self._v[_n] = value
                return self
            return setter
        raise AttributeError(name)

    def get(self, name):
        return self._v.get(name)


class SeqKitConfig(_Builder):
    """bigseqkit/helper.go:25-27,105-138"""
    _fields = tuple(_CONFIG_FIELDS)

    def to_dict(self):
        return {k: self._v.get(k) for k in _CONFIG_FIELDS}


class _CmdOptions(_Builder):
    op = ""

    def __init__(self, **kwargs):
        super().__init__()
        object.__setattr__(self, "_cfg", SeqKitConfig())
        # kwargs sugar of the Python wrapper (bigseqkit-py/bigseqkit/helper.py:34-43):
        # name.capitalize() is looked up on the option struct, then on Config
        for k, v in kwargs.items():
            name = k[0].upper() + k[1:]
            if name in self._fields:
                self._v[name] = v
            elif name in _CONFIG_FIELDS:
                self._cfg._v[name] = v
            else:
                raise KeyError(f"unknown option {k}")

    def Config(self, cfg):
        object.__setattr__(self, "_cfg", cfg)
        return self

    def to_json(self):
        d = {"Config": self._cfg.to_dict()}
        for k in self._fields:
            d[k] = self._v.get(k)
        return json.dumps(d, separators=(",", ":")) + "\n"


def _make(op):
    return type("SeqKit" + op + "Options", (_CmdOptions,), {"op": op, "_fields": tuple(_FIELDS[op])})


SeqKitStatsOptions = _make("Stats")
SeqKitSeqOptions = type("SeqKitSeqOptions", (_CmdOptions,), {"op": "SeqTransform", "_fields": tuple(_FIELDS["SeqTransform"])})
SeqKitGrepOptions = _make("Grep")
SeqKitLocateOptions = _make("Locate")
SeqKitSubseqOptions = type("SeqKitSubseqOptions", (_CmdOptions,), {"op": "SubseqTransform", "_fields": tuple(_FIELDS["SubseqTransform"])})
SeqKitTranslateOptions = _make("Translate")
SeqKitRmDupOptions = _make("RmDup")
SeqKitFq2FaOptions = _make("Fq2Fa")
SeqKitRangeOptions = _make("Range")
SeqKitHeadOptions = _make("Head")
SeqKitDuplicateOptions = _make("Duplicate")
SeqKitRenameOptions = _make("Rename")
SeqKitSortOptions = _make("Sort")
SeqKitFaidxOptions = _make("Faidx")
SeqKitPairOptions = _make("Pair")
SeqKitCommonOptions = _make("Common")
SeqKitConcatOptions = _make("Concat")
