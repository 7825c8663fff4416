"""Result files of the two kernels -- SURVEY.md section 8(f) rank 4.

The reference closes a search by writing one JSON file per mask MIP id (and optionally one per matched MIP id) that the
rest of its pipeline (gradient scoring, normalisation, export) reads back:

  * `JSONNeuronMatchesWriter.write / writeMatchesByMask / writeMatchesByTarget`
    (PERSIST/dataio/fs/JSONNeuronMatchesWriter.java:44-90) on top of
    `MatchEntitiesGrouping.groupByMaskFields / groupByTargetFields` (API/results/MatchEntitiesGrouping.java:54-148) and
    `ItemsHandling.groupItems` (API/results/ItemsHandling.java:36-70);
  * file = `<dir>/<mipId>.json` = `{"inputImage": <entity>, "results": [<match>, ...]}`
    (API/results/GroupedMatchedEntities.java:12-33, PERSIST/dataio/fileutils/ItemsWriterToJSONFile.java:46-64);
  * `JSONNeuronMatchesReader.readMatchesByMask` -> `MatchEntitiesGrouping.expandResultsByMask` (:150-173) is the way back.

Entities are the Jackson documents themselves: a neuron entity is a dict with the reference's property names (`class`,
`mipId`, `libraryName`, `alignmentSpace`, `publishedName`, `computeFiles`, ...; API/model/AbstractNeuronEntity.java:24-50)
and a match a dict with `maskImage`, `image`, `mirrored`, `matchComputeFiles`, `normalizedScore`, `matchingPixels`,
`matchingPixelsRatio`, `gradientAreaGap`, `highExpressionArea` (API/model/AbstractMatchEntity.java:22-31,
CDMatchEntity.java:12-20).  `match_entities` builds them from the processor's `CDMatch` records / `MatchTable` columns.

What is restated of the Java behaviour:
  - grouping key = the selected field of the mask (the MIP id in every command: ColorDepthSearchCmd.java:403-409,
    CalculateGradientScoresCmd.java:443-449); the group's `inputImage` is the FIRST match's mask entity, duplicated,
    without its InputColorDepthImage / GradientImage / ZGapImage compute files -- those move to every result's
    `matchComputeFiles` as MaskColorDepthImage / MaskGradientImage / MaskZGapImage;
  - a duplicated match carries only the fields `duplicate()` copies (CDMatchEntity.java:140-160,
    AbstractMatchEntity.java:151-157): no entity id, no tags; a duplicated neuron keeps what `copyFrom` copies
    (AbstractNeuronEntity.java:317-331 and the EM/LM subclasses) -- no entity id, no dates;
  - results whose matched image is missing are dropped, the rest are sorted (stable) with the writer's ordering;
  - `@JsonInclude(NON_EMPTY)` (AbstractBaseEntity.java:19): nulls, empty strings, empty maps and collections are not
    written; `float` scores are written with the shortest decimal that reads back to the same float32.
"""
import json
import os
import re
import time
from typing import Callable, Dict, Iterable, List, Optional, Sequence

import numpy as np

# ComputeFileType -> MatchComputeFileType (MatchEntitiesGrouping.java:65-70)
_MASK_COMPUTE_FILES = (("InputColorDepthImage", "MaskColorDepthImage"),
                       ("GradientImage", "MaskGradientImage"),
                       ("ZGapImage", "MaskZGapImage"))
# what AbstractNeuronEntity.copyFrom and the EM / LM duplicate() carry over
_NEURON_FIELDS = ("class", "mipId", "alignmentSpace", "libraryName", "publishedName", "sourceRefId", "neuronTerms",
                  "computeFiles", "tags", "validationErrors", "processedTags", "datasetLabels",
                  "neuronType", "neuronInstance",                                                       # EMNeuronEntity
                  "internalLineName", "slideCode", "anatomicalArea", "gender", "objective", "notStaged", "publishError")  # LM
# what CDMatchEntity.duplicate copies (matchFound and errors are @JsonIgnore)
_MATCH_FIELDS = ("class", "mirrored", "matchComputeFiles", "normalizedScore", "matchingPixels", "matchingPixelsRatio",
                 "bidirectionalAreaGap", "gradientAreaGap", "highExpressionArea")


def _shortest_float32(v) -> float:
    """A Python float whose repr is the shortest decimal identifying float32(v) (what Float.toString / Jackson print)."""
    return float(np.format_float_positional(np.float32(v), unique=True, trim="0"))


def _is_empty(v) -> bool:
    return v is None or (isinstance(v, (str, list, tuple, dict, set, frozenset)) and len(v) == 0)


def _clean(doc):
    """NON_EMPTY inclusion, recursively; numpy scalars become JSON numbers."""
    if isinstance(doc, dict):
        out = {}
        for k, v in doc.items():
            v = _clean(v)
            if not _is_empty(v):
                out[k] = v
        return out
    if isinstance(doc, (list, tuple)):
        return [_clean(v) for v in doc]
    if isinstance(doc, (set, frozenset)):
        return sorted(_clean(v) for v in doc)
    if isinstance(doc, np.floating):
        return _shortest_float32(doc) if doc.dtype == np.float32 else float(doc)
    if isinstance(doc, np.integer):
        return int(doc)
    if isinstance(doc, np.bool_):
        return bool(doc)
    return doc


def duplicate_neuron(entity: Optional[dict]) -> Optional[dict]:
    """EMNeuronEntity / LMNeuronEntity.duplicate(): a new entity with the copied attributes, no entity id or dates."""
    if entity is None:
        return None
    out = {k: entity[k] for k in _NEURON_FIELDS if k in entity}
    if "computeFiles" in out:
        out["computeFiles"] = dict(out["computeFiles"])
    return out


def _known_neuron(entity: Optional[dict]) -> Optional[dict]:
    """An entity as Jackson would write it back: the declared properties only (unknown ones are ignored on read)."""
    if entity is None:
        return None
    return {k: entity[k] for k in ("entityId", "createdDate", "updatedDate") + _NEURON_FIELDS if k in entity}


def _duplicate_match(match: dict) -> dict:
    out = {k: match[k] for k in _MATCH_FIELDS if k in match}
    out["matchComputeFiles"] = dict(match.get("matchComputeFiles") or {})
    out.setdefault("mirrored", False)
    return out


def _regroup(matches: Sequence[dict], key_side: str, other_side: str, selector: Callable[[dict], str],
             ordering: Optional[Callable[[dict], float]]):
    """groupByMaskFields (key_side = "maskImage") / groupByTargetFields (key_side = "image"): a list of
    {"inputImage", "results"} in first-appearance order of the key."""
    groups: Dict[str, dict] = {}
    for m in matches:
        key_entity = m.get(key_side)
        if key_entity is None:
            raise ValueError("a match without its %s cannot be grouped" % key_side)
        dup = _duplicate_match(m)
        for cft, mcft in _MASK_COMPUTE_FILES:   # setMatchComputeFileData: a missing file removes the entry
            fd = (key_entity.get("computeFiles") or {}).get(cft)
            if fd is not None:
                dup["matchComputeFiles"][mcft] = fd
            else:
                dup["matchComputeFiles"].pop(mcft, None)
        dup["image"] = _known_neuron(m.get(other_side))
        now = int(time.time() * 1000)            # a duplicate is a new entity: its dates are those of the write
        dup["createdDate"] = now
        dup["updatedDate"] = now
        k = selector(key_entity)
        g = groups.get(k)
        if g is None:
            key_dup = duplicate_neuron(key_entity)
            cf = key_dup.get("computeFiles") or {}
            for cft, _ in _MASK_COMPUTE_FILES:
                cf.pop(cft, None)
            key_dup["computeFiles"] = cf
            g = groups[k] = {"inputImage": key_dup, "results": []}
        if dup["image"] is not None:             # the filter m -> m.getMatchedImage() != null
            g["results"].append(dup)
    out = list(groups.values())
    if ordering is not None:
        for g in out:
            g["results"].sort(key=ordering)      # List.sort is stable, so is this
    return out


def group_by_mask(matches, selector=lambda e: e.get("mipId"), ordering=None):
    return _regroup(matches, "maskImage", "image", selector, ordering)


def group_by_target(matches, selector=lambda e: e.get("mipId"), ordering=None):
    return _regroup(matches, "image", "maskImage", selector, ordering)


def expand_results_by_mask(grouped: dict) -> List[dict]:
    """MatchEntitiesGrouping.expandResultsByMask (:150-173): each result becomes a full match again, the mask's three
    compute files restored from the result's matchComputeFiles, which is then emptied."""
    out = []
    for r in grouped.get("results") or []:
        mask = duplicate_neuron(grouped.get("inputImage")) or {}
        cf = dict(mask.get("computeFiles") or {})
        mcf = r.get("matchComputeFiles") or {}
        for cft, mcft in _MASK_COMPUTE_FILES:
            if mcf.get(mcft) is not None:
                cf[cft] = mcf[mcft]
            else:
                cf.pop(cft, None)
        mask["computeFiles"] = cf
        m = _duplicate_match(r)
        m["matchComputeFiles"] = {}
        m["maskImage"] = mask
        m["image"] = r.get("image")
        out.append(m)
    return out


def expand_results_by_target(grouped: dict) -> List[dict]:
    """MatchEntitiesGrouping.expandResultsByTarget (:175-198): as above with the two sides swapped back."""
    out = []
    for m in expand_results_by_mask(grouped):
        m["maskImage"], m["image"] = m["image"], m["maskImage"]
        out.append(m)
    return out


# ---- orderings used by the reference's commands --------------------------------------------------------------------
def by_matching_pixels_desc(m: dict) -> float:      # ColorDepthSearchCmd.java:406
    return -float(m["matchingPixels"])


def by_normalized_score_desc(m: dict) -> float:     # CalculateGradientScoresCmd.java:446
    return -float(m["normalizedScore"])


class JSONNeuronMatchesWriter:
    """Counterpart of PERSIST/dataio/fs/JSONNeuronMatchesWriter.java: `write(matches)` groups by the mask (and by the
    matched image when `perMatchesOutputDir` is given) and writes one `<mipId>.json` per group; returns the number of files."""

    def __init__(self, resultsGrouping: Callable[[dict], str] = lambda e: e.get("mipId"),
                 matchOrdering: Optional[Callable[[dict], float]] = by_matching_pixels_desc,
                 perMasksOutputDir: Optional[str] = None, perMatchesOutputDir: Optional[str] = None, prettyPrint: bool = True):
        self.resultsGrouping = resultsGrouping
        self.matchOrdering = matchOrdering
        self.perMasksOutputDir = perMasksOutputDir
        self.perMatchesOutputDir = perMatchesOutputDir
        self.prettyPrint = prettyPrint
        self.written: List[str] = []             # paths of the files written by the last call

    def write(self, matches: Sequence[dict]) -> int:
        self.written = []
        n = 0
        if self.perMasksOutputDir is not None:
            n += self._write_groups(group_by_mask(matches, self.resultsGrouping, self.matchOrdering), self.perMasksOutputDir)
        if self.perMatchesOutputDir is not None:
            n += self._write_groups(group_by_target(matches, self.resultsGrouping, self.matchOrdering), self.perMatchesOutputDir)
        return n

    def writeUpdates(self, matches: Sequence[dict], fieldSelectors=None) -> int:   # :58-60: rewrites the per-mask files
        self.written = []
        return self._write_groups(group_by_mask(matches, self.resultsGrouping, self.matchOrdering), self.perMasksOutputDir)

    def _write_groups(self, groups, out_dir) -> int:
        if not groups:
            return 0
        os.makedirs(out_dir, exist_ok=True)
        for g in groups:
            name = self.resultsGrouping(g["inputImage"])
            if name is None or not str(name).strip():
                continue                             # ItemsWriterToJSONFile.getJsonFile: blank name -> nothing written
            # the name comes from input metadata: keep it a plain file name inside out_dir
            name = re.sub(r"[^A-Za-z0-9._-]", "_", os.path.basename(str(name))).lstrip(".") or "_"
            path = os.path.join(out_dir, name + ".json")
            tmp = path + ".tmp%d" % os.getpid()
            with open(tmp, "w", encoding="utf-8") as f:
                json.dump(_clean(g), f, indent=2 if self.prettyPrint else None,
                          separators=None if self.prettyPrint else (",", ":"), ensure_ascii=False)
            os.replace(tmp, path)
            self.written.append(path)
        return len(groups)


class JSONNeuronMatchesReader:
    """Counterpart of PERSIST/dataio/fs/JSONNeuronMatchesReader.java: the per-mask / per-target files back to matches."""

    @staticmethod
    def _load(path):
        if not path.endswith(".json") and not os.path.exists(path):
            path += ".json"
        with open(path, "r", encoding="utf-8") as f:
            return json.load(f)

    def readMatchesByMask(self, location: str, maskMipIds: Iterable[str]) -> List[dict]:
        out = []
        for mid in maskMipIds:
            out += expand_results_by_mask(self._load(os.path.join(location, mid)))
        return out

    def readMatchesByTarget(self, location: str, targetMipIds: Iterable[str]) -> List[dict]:
        out = []
        for mid in targetMipIds:
            out += expand_results_by_target(self._load(os.path.join(location, mid)))
        return out


def read_matches_list(path: str) -> List[dict]:
    """A plain JSON array of matches (the form of the reference's test resource cdsmatches/testcdsmatches.json)."""
    with open(path, "r", encoding="utf-8") as f:
        return json.load(f)


def match_entities(matches, mask_entities: Sequence[dict], target_entities: Sequence[dict]) -> List[dict]:
    """CDMatch records or a MatchTable (colormipsearch_amd.processor) -> match documents.  `mask_entities[i]` /
    `target_entities[j]` are the neuron documents of mask i / library target j.  Scores that were not computed (None or
    -1 in a table) are left out, as a null Long is."""
    if hasattr(matches, "to_matches"):
        matches = matches.to_matches()
    out = []
    for mt in matches:
        d = {"class": "org.janelia.colormipsearch.model.CDMatchEntity",
             "maskImage": mask_entities[mt.maskIndex], "image": target_entities[mt.targetIndex],
             "mirrored": bool(mt.mirrored), "matchComputeFiles": {},
             "matchingPixels": int(mt.matchingPixels), "matchingPixelsRatio": np.float32(mt.matchingPixelsRatio)}
        if mt.normalizedScore is not None:
            d["normalizedScore"] = np.float32(mt.normalizedScore)
        if mt.gradientAreaGap is not None and mt.gradientAreaGap >= 0:
            d["gradientAreaGap"] = int(mt.gradientAreaGap)
        if mt.highExpressionArea is not None and mt.highExpressionArea >= 0:
            d["highExpressionArea"] = int(mt.highExpressionArea)
        out.append(d)
    return out


# ---- the input side: where an entity's images are ----------------------------------------------------------------------------------
def file_data(value):
    """A `computeFiles` value as the image loaders of this package take it.  The reference writes a FileData as a plain string when it is
    a file and as {"dataType": "zipEntry", "fileName": <archive>, "entryName": <entry>} when it is an entry of a zip archive
    (API/model/json/FileDataSerializer.java:13-26) and reads both, plus {"dataType": "file", "fileName": <dir>, "entryName": <name>} =
    the file <dir>/<name> (FileDataDeserializer.java:14-36, FileData.fromComponentsUsingParentPath :60-74).
    -> a path, a pair (archive, entry) -- `mipio.read_mip`, `cds.native_path` and with it the native batch loaders take either --, or None
    for null / blank."""
    if value is None:
        return None
    if isinstance(value, str):
        return value if value.strip() else None
    if isinstance(value, dict):
        name, entry = value.get("fileName"), value.get("entryName")
        if value.get("dataType", "file") == "zipEntry":
            return (name, entry) if name and entry else None
        if not name:
            return None
        return os.path.join(name, entry) if entry else name
    raise ValueError(f"not a FileData: {value!r}")


def compute_file(entity: Optional[dict], compute_file_type: str):
    """AbstractNeuronEntity.getComputeFileData (API/model/AbstractNeuronEntity.java): the entity's file of that ComputeFileType, or None"""
    return file_data(((entity or {}).get("computeFiles") or {}).get(compute_file_type))


def variant_files(target_entities: Sequence[dict]):
    """(target files, gradient files, zgap files) of a list of neuron entities: what CalculateGradientScoresCmd's variant suppliers open
    per match (cmd/CalculateGradientScoresCmd.java:541-586: InputColorDepthImage, GradientImage, ZGapImage) as the three lists
    `processor.calculateGradientScoresFromFilesTable`, `cds.Library.load_files` and `cds.VariantShardWriter.append_files` take."""
    return ([compute_file(e, "InputColorDepthImage") for e in target_entities],
            [compute_file(e, "GradientImage") for e in target_entities],
            [compute_file(e, "ZGapImage") for e in target_entities])
