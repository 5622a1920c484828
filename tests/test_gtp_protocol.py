"""CPU: the GTP protocol layer (agogo_amd/host/gtp.hpp Protocol) against the reference's own known-answer strings
(internal/gtp/gtp_test.go:9-31, Test_General: engine New(nil, "xx", "1", nil)), plus parsing edge cases of gtp.go:83-112."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_proto(script, name="xx", version="1"):
    exe = os.path.join(ROOT, "tests", "cpp", "gtp_main")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", ROOT, "tests/cpp/gtp_main"])
    out = subprocess.run([exe, "proto", name, version], input=script, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    return out.stdout


def test_reference_known_answers():
    # the four reply strings gtp_test.go pins, byte for byte (including the blank line that ends a reply)
    assert run_proto("version\n") == "= 1\n\n"
    assert run_proto("known_command hello\n") == "= false\n\n"
    assert run_proto("known_command name\n") == "= true\n\n"
    assert run_proto("completelyUnheardOfCommand xxx\n") == '? Unknown command "completelyunheardofcommand"\n\n'


def test_parsing_edges():
    # lower-casing and trimming (gtp.go:110-112), optional id, an id alone is ignored (gtp.go:96-98), comments, blank lines
    out = run_proto("  NAME  \n42\n\n# only a comment\n9 Protocol_Version # trailing\nlist_commands\nplay b a1\nquit\nname\n")
    replies = out.split("\n\n")
    assert replies[0] == "= xx"
    assert replies[1] == "=9 2"
    assert replies[2].startswith("= protocol_version\nname\nversion\nknown_command\nlist_commands\nquit\nboardsize")
    assert replies[3] == "? no game attached"      # board commands need a game (the reference dereferences its nil game here)
    assert replies[4] == "="                       # quit
    assert replies[5] == ""                        # nothing is answered after quit
