"""Partition lines and model strings: the known answers of the reference's
test/src/msa.cpp:17-245 ("msa_t parse partition line") and :247-284 ("msa_t
partition datafile"), section for section, through the C ABI."""
import os

import pytest

import root_digger_amd as rd
import util

TWO = [(123, 4123), (5122, 12411)]


def test_plain_lines():                                  # test/src/msa.cpp:18-43
    for line in ("DNA, PART_0 = 123-4123", "DNA,PART_0=123-4123"):
        pi = rd.parse_partition_info(line)
        assert (pi["model_name"], pi["partition_name"], pi["parts"]) == ("DNA", "PART_0", [(123, 4123)])
    pi = rd.parse_partition_info("DNA,PART_0=123-4123, 5122-12411")
    assert pi["parts"] == TWO


@pytest.mark.parametrize("model,freq", [                 # :45-93
    ("DNA+F", "emperical"), ("DNA+FO", "estimate"), ("DNA+FE", "equal"),
    ("DNA+FU{0.25/0.25/0.25/0.25}", "user")])
def test_frequency_options(model, freq):
    pi = rd.parse_partition_info(model + ",PART_0=123-4123, 5122-12411")
    assert pi["model_name"] == model and pi["subst_str"] == "DNA"
    assert pi["partition_name"] == "PART_0" and pi["parts"] == TWO
    assert pi["freq_type"] == freq


@pytest.mark.parametrize("model,kind,prop", [            # :95-130
    ("DNA+I", "estimate", None), ("DNA+IC", "emperical", None), ("DNA+IU{0.25}", "user", 0.25)])
def test_invariant_options(model, kind, prop):
    pi = rd.parse_partition_info(model + ",PART_0=123-4123, 5122-12411")
    assert pi["model_name"] == model and pi["parts"] == TWO
    assert pi["invar"]["type"] == kind
    if prop is not None:
        assert pi["invar"]["user_prop"] == prop


@pytest.mark.parametrize("model,kind,cats,category,alpha", [   # :131-207
    ("DNA+G", "estimate", 4, "MEAN", None),
    ("DNA+G2", "estimate", 2, "MEAN", None),
    ("DNA+G2{0.25}", "user", 2, "MEAN", 0.25),
    ("DNA+GA", "estimate", 4, "MEDIAN", None),
    ("DNA+R4", "estimate", 4, "FREE", None),
    ("DNA+R2{0.2/0.2}{0.1/0.1}", "estimate", 2, "FREE", None)])
def test_rate_heterogeneity_options(model, kind, cats, category, alpha):
    pi = rd.parse_partition_info(model + ",PART_0=123-4123, 5122-12411")
    assert pi["model_name"] == model and pi["partition_name"] == "PART_0" and pi["parts"] == TWO
    r = pi["ratehet"]
    assert (r["type"], r["rate_cats"], r["rate_category_type"]) == (kind, cats, category)
    if alpha is not None:
        assert r["alpha"] == alpha and r["alpha_init"]


def test_all_together():                                  # :209-224
    pi = rd.parse_partition_info("DNA+G2{0.25}+F+I,PART_0=123-4123, 5122-12411")
    assert pi["model_name"] == "DNA+G2{0.25}+F+I" and pi["parts"] == TWO
    assert pi["ratehet"]["type"] == "user" and pi["ratehet"]["rate_cats"] == 2
    assert pi["ratehet"]["alpha"] == 0.25
    assert pi["invar"]["type"] == "estimate" and pi["freq_type"] == "emperical"


@pytest.mark.parametrize("line", [                        # :226-245
    "DNA PART_0 = 123-4123", "DNA, PART_0  123-4123", "DNA, PART_0 = 1234123",
    "DNA, PART_0 = 123=4123", ", PART_0 = 123-4123"])
def test_malformed_lines_throw(line):
    with pytest.raises(rd.RdamdError):
        rd.parse_partition_info(line)


def test_model_strings_beyond_the_reference_tests():
    mi = rd.parse_model_info("UNREST+G4+ASC_S{0.1/0.2/0.3}+M")
    assert mi["subst_str"] == "UNREST" and mi["ratehet"]["rate_cats"] == 4
    assert mi["asc"] == {"type": "stam", "fels_weight": 0.0, "stam_weights": [0.1, 0.2, 0.3]}
    assert rd.parse_model_info("BIN+ASC_F{2.5}")["asc"]["fels_weight"] == 2.5
    assert rd.parse_model_info("unrest")["ratehet"]["rate_cats"] == 0      # no +G: main() turns 0 into 1
    assert rd.parse_partition_info("DNA, p = 5, 10-20")["parts"] == [(5, 5), (10, 20)]
    for bad in ("DNA+", "DNA+Q", "DNA+IU{x}", "+G"):
        with pytest.raises(rd.RdamdError):
            rd.parse_model_info(bad)
    with pytest.raises(rd.RdamdError):
        rd.parse_partition_info("DNA, p = 20-10")


@pytest.mark.parametrize("lines,lengths", [               # test/src/msa.cpp:247-284
    (["DNA, PART_0 = 1-100"], [100]),
    (["DNA, PART_0 = 1-100, 200-300"], [201]),
    (["DNA, PART_0 = 1-100", "DNA, PART_1 = 200-300"], [100, 101]),
    (["DNA, PART_0 = 1-100, 500-520", "DNA, PART_1 = 200-300, 400-500"], [121, 202])])
def test_partitioned_datafile(lines, lengths):
    phy = os.path.join(util.DATA, "101.phy")
    raw = rd.msa_partition_probe(phy, lines, compress=False)
    assert [n for n, _ in raw] == lengths and [w for _, w in raw] == lengths
    packed = rd.msa_partition_probe(phy, lines, compress=True)
    for (n, w), full in zip(packed, lengths):              # compression keeps the total weight
        assert n <= full and w == full
    with pytest.raises(rd.RdamdError):
        rd.msa_partition_probe(phy, ["DNA, P = 0-10"])
    with pytest.raises(rd.RdamdError):
        rd.msa_partition_probe(phy, ["DNA, P = 1-100000000"])
