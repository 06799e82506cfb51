"""The binding INTEGRATION.md shows (section 1 struct, section 2 upload, section 3 evaluation hook, section 5 / 5.1 table members) is
COMPILED here against the reference's own headers (x264_t, x264_frame_t, x264_weight_t, x264_mc_functions_t) and include/x264hip.h:
the C code blocks are taken out of the document as they stand, `h->hip` -- a member the patch would add to x264_t -- is redirected to
a file-scope x264_hip_t, and the per-frame handle `hip_slot` to an existing int member.  Syntax and types only (-fsyntax-only);
nothing of the reference is copied or linked.  Needs /root/reference and the generated config.h of oracle/_ref (build container)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
CFG = os.path.join(ROOT, "oracle", "_ref", "cfg")

pytestmark = pytest.mark.skipif(not (os.path.isdir(REF) and os.path.exists(os.path.join(CFG, "config.h"))),
                                reason="needs the reference headers and oracle/_ref/cfg/config.h (build container)")


def _blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return re.findall(r"```c\n(.*?)```", text, re.S)


def _pick(blocks, marker):
    got = [b for b in blocks if marker in b]
    assert len(got) == 1, (marker, len(got))
    return got[0]


@pytest.mark.parametrize("depth", [8, 10])
def test_integration_snippets_compile_against_reference_headers(depth, tmp_path):
    blocks = _blocks()
    struct = _pick(blocks, "} x264_hip_t;")
    upload = _pick(blocks, "h->hip.frame_put( h->hip.ctx, fenc->hip_slot")
    hook = _pick(blocks, "#if HAVE_HIPLOOKAHEAD")
    hpel = _pick(blocks, "h->hip.hpel_filter( ctx")
    tables = _pick(blocks, "h->hip.mc_fill( h->hip.ctx, &f )")
    bind = _pick(blocks, "h->hip.mc_bind_handle( h->hip.ctx, h );")
    fillers = _pick(blocks, "x264hip_pixel_fill( ctx, &pf )")
    redirect = lambda s: s.replace("h->hip.", "g_hip.")  # noqa: E731
    src = """
#include "common/common.h"
#define HAVE_HIPLOOKAHEAD 1
#define hip_slot i_frame            /* the patch adds `int hip_slot` to x264_frame_t: any int member type-checks the same */
%s
static x264_hip_t g_hip;            /* the patch adds `x264_hip_t hip` to x264_t */
static void hip_check( x264_t *h, int rc ) { if( rc ) g_hip.b_fatal_error = 1; (void)h; }

void snippet_upload( x264_t *h, x264_frame_t *fenc )
{
%s
}

int snippet_hook( x264_t *h, x264_frame_t **frames, int p0, int p1, int b, const x264_weight_t *w, int do_search[2] )
{
    x264_frame_t *fenc = frames[b];
    int i_score = 0;
%s
    { i_score = -1; }
    return i_score;
}

void snippet_hpel( x264_t *h, x264_frame_t *frame, int p, int offs, intptr_t stride, int width, int height, int start )
{
    x264hip_ctx *ctx = g_hip.ctx;
    (void)h;
%s
}

void snippet_tables( x264_t *h )
{
%s
%s
}

void snippet_fillers( x264_t *h, x264hip_ctx *ctx )
{
%s
}
""" % (struct, redirect(upload), redirect(hook), redirect(hpel), redirect(tables), redirect(bind), fillers.replace("/* ... the other eight */", "").replace("/* ... the other four  */", ""))
    tu = tmp_path / "integration_snippets.c"
    tu.write_text(src)
    cmd = ["gcc", "-std=gnu99", "-fsyntax-only", "-Wall", "-Werror=implicit-function-declaration", "-Werror=incompatible-pointer-types",
           "-Werror=int-conversion", "-DHIGH_BIT_DEPTH=%d" % (depth > 8), "-DBIT_DEPTH=%d" % depth, "-I" + CFG, "-I" + REF,
           "-I" + os.path.join(ROOT, "include"), str(tu)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
