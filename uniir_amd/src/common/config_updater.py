"""In-place YAML tweak used by the shell drivers (drop-in for UniIR src/common/config_updater.py :25-42):
--update_mbeir_yaml_instruct_status sets experiment.instruct_status ("Instruct" / "NoInstruct") and
data_config.enable_query_instruct, leaving every other line (comments, ${...} references) untouched."""
import argparse
import re


def update_mbeir_yaml_instruct_status(path, enable_instruct):
    status = "Instruct" if enable_instruct else "NoInstruct"
    out, section = [], None
    with open(path, "r") as f:
        for line in f:
            m = re.match(r"^([A-Za-z_][\w]*):", line)
            if m:
                section = m.group(1)
            if section == "experiment" and re.match(r"^\s+instruct_status:", line):
                line = re.sub(r"(instruct_status:\s*).*", rf'\g<1>"{status}"', line.rstrip("\n")) + "\n"
            if section == "data_config" and re.match(r"^\s+enable_query_instruct:", line):
                line = re.sub(r"(enable_query_instruct:\s*).*", rf"\g<1>{'True' if enable_instruct else 'False'}",
                              line.rstrip("\n")) + "\n"
            out.append(line)
    with open(path, "w") as f:
        f.writelines(out)


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--update_mbeir_yaml_instruct_status", action="store_true")
    p.add_argument("--mbeir_yaml_file_path", type=str)
    p.add_argument("--enable_instruct", type=str, default="True")
    a = p.parse_args()
    if a.update_mbeir_yaml_instruct_status:
        update_mbeir_yaml_instruct_status(a.mbeir_yaml_file_path, str(a.enable_instruct).lower() in ("1", "true", "yes"))
