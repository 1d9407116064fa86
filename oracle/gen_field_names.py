"""Test infrastructure (authoring container only: reads /root/reference): the field names of the reference's request / config
dataclasses and the request attributes its Qwen-Image pipelines read, written to tests/golden/reference_field_names.json.
`tests/test_host_logic.py::test_request_and_config_accept_the_reference_field_names` checks this build's dataclasses against
the committed file (the GPU box has no /root/reference)."""
import json
import os
import re

REF = "/root/reference/vllm_omni/diffusion"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_field_names.json")


def class_fields(src: str, cls: str) -> list[str]:
    body = re.search(r"class " + cls + r"\b.*?:\n(.*?)(?=\nclass |\n@dataclass|\Z)", src, re.S).group(1)
    body = body.split("\n    def ")[0]                      # fields precede the first method
    return re.findall(r"^\s{4}(\w+)\s*:\s*[^=\n]+(?:=.*)?$", body, re.M)


def main() -> None:
    data = open(os.path.join(REF, "data.py")).read()
    req = open(os.path.join(REF, "request.py")).read()
    read = set()
    for f in ("pipeline_qwen_image.py", "pipeline_qwen_image_edit.py", "pipeline_qwen_image_edit_plus.py",
              "pipeline_qwen_image_layered.py"):
        read |= set(re.findall(r"\breq\.(\w+)", open(os.path.join(REF, "models", "qwen_image", f)).read()))
    req_fields = class_fields(req, "OmniDiffusionRequest")
    out = {"source": "vllm_omni/diffusion/{data,request}.py, models/qwen_image/pipeline_qwen_image*.py",
           "OmniDiffusionConfig": class_fields(data, "OmniDiffusionConfig"),
           "DiffusionParallelConfig": class_fields(data, "DiffusionParallelConfig"),
           "OmniDiffusionRequest": req_fields,
           "request_fields_read_by_the_qwen_image_pipelines": sorted(read & set(req_fields)),
           "request_attributes_set_by_their_pre_process_steps": sorted(read - set(req_fields))}
    with open(OUT, "w") as fh:
        json.dump(out, fh, indent=1)
    print(OUT, {k: len(v) for k, v in out.items() if isinstance(v, list)})


if __name__ == "__main__":
    main()
