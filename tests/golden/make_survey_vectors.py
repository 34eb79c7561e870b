"""Encodes the reference outputs quoted in SURVEY.md §4.2-§4.4 / §7.3 as JSON.
Data only: every string below is an expected output of the reference, keyed by
the input files and query it belongs to."""
import json

HDR = "Pos\tRef\tAlt\tSamples\n"


def rows(*r):
    return HDR + "".join(x + "\n" for x in r)


vectors = {
    "readme": {  # reference README.md:51-62, 89-96
        "x": {"num_mutations": 75, "num_mutations_samples": 75, "vertices": 212, "edges": 287, "seq_length": 1074,
              "classes": 2, "t6_10_105": 8},
    },
    "G1": {"fasta": "x.fa", "vcf": "x.vcf", "type": 6, "region": [10, 105],
           "text": rows("10\tC\tT\t1(1|1) ", "14\tG\tA\t1(1|0) ", "34\tT\tA\t1(1|1) ", "39\tT\tA\t1(1|0) ",
                        "52\tT\tG\t1(1|0) ", "58\t\tT\t1(0|1) ", "100\tT\tC\t1(1|1) ", "103\tT\tC\t1(1|0) ")},
    "G1_t4": {"fasta": "x.fa", "vcf": "x.vcf", "type": 4, "sample": "1", "region": [10, 105],
              "text": rows("10\tC\tT\t1(1|1) ", "14\tG\tA\t1(1|0) ", "34\tT\tA\t1(1|1) ", "39\tT\tA\t1(1|0) ",
                           "52\tT\tG\t1(1|0) ", "58\t\tT\t1(0|1) ", "100\tT\tC\t1(1|1) ", "103\tT\tC\t1(1|0) ")},
    "G2": {"fasta": "x.fa", "vcf": "x.vcf", "type": 6, "region": [1, 1001], "count": 75,
           "contains": ["58\t\tT\t", "172\t\tA\t", "345\t\tTGA\t", "500\t\tA\t", "553\t\tG\t", "668\t\tA\t",
                        "681\t\tT\t", "860\t\tA\t", "467\tC\t\t", "670\tG\t\t", "790\tG\t\t", "940\tC\t\t",
                        "973\tGG\t\t", "272\tTA\tCG\t"],
           "last_row": "1000\tG\tA\t1(0|1) "},
    "G3": {"fasta": "x.small.fa", "vcf": "x.small.vcf", "type": 6, "region": [1, 80],
           "stats": {"vertices": 18, "edges": 27, "seq_length": 88, "classes": 2},
           "text": rows("9\tG\tA\t1(1|0) ", "10\tC\tT\t1(1|0) ", "10\tC\tA\t1(1|0) ", "10\tC\tAAA\t1(1|0) ",
                        "25\t\tT\t1(0|1) ", "26\t\tA\t1(0|1) ", "39\tT\t\t1(0|1) ", "41\tC\t\t1(0|1) ",
                        "55\tC\t\t1(1|1) ")},
    "G3_t4": {"fasta": "x.small.fa", "vcf": "x.small.vcf", "type": 4, "sample": "1", "region": [1, 80],
              "text": rows("9\tG\tA\t1(1|0) ", "10\tC\tT\t1(1|0) ", "25\t\tT\t1(0|1) ", "26\t\tA\t1(0|1) ",
                           "39\tT\t\t1(0|1) ", "41\tC\t\t1(0|1) ", "55\tC\t\t1(1|1) ")},
    "G4": {"fasta": "x.small.fa", "vcf": "g4.vcf", "type": 6, "region": [1, 80],
           "stats": {"vertices": 12, "edges": 18, "seq_length": 85, "classes": 4},
           "text": rows("9\tG\tA\tS2(1/1) S10(0|1) S1(1|0) ", "20\t\tC\tS2(0|1) S10(1|0) ",
                        "20\tT\tG\tS2(0|1) S10(1|0) ", "39\tT\t\tS10(0|1) S1(1|0) ", "54\t\tAG\tS2(0|1) ")},
    "G4_probes": {"fasta": "x.small.fa", "vcf": "g4.vcf", "type": 6, "probes": [
        {"region": [19, 30], "text": rows("20\t\tC\tS2(0|1) S10(1|0) ", "20\tT\tG\tS2(0|1) S10(1|0) "), "early_out": False},
        {"region": [20, 30], "text": rows("20\tT\tG\tS2(0|1) S10(1|0) "), "early_out": False},
        {"region": [9, 12], "text": rows(), "early_out": False},
        {"region": [21, 30], "text": rows(), "early_out": True},
        {"region": [10, 12], "text": rows(), "early_out": True}]},
    "x_small_graph": {  # vertex -> [off, len, class, ref index or 0, out-neighbours in query-time order]
        "fasta": "x.small.fa", "vcf": "x.small.vcf",
        "vertices": {"0": [0, 8, 0, 1, [1, 3]], "1": [8, 1, 0, 9, [2]], "2": [9, 0, 0, 10, [4, 6, 7, 8]],
                     "3": [80, 1, 1, 0, [2]], "4": [9, 1, 0, 10, [5]], "5": [10, 15, 0, 11, [9, 10]],
                     "6": [81, 1, 1, 0, [5]], "7": [82, 1, 1, 0, [5]], "8": [83, 3, 1, 0, [5]],
                     "9": [25, 1, 0, 26, [11, 12]], "10": [86, 1, 1, 0, [9]], "11": [26, 12, 0, 27, [13, 14]],
                     "12": [87, 1, 1, 0, [11]], "13": [38, 1, 0, 39, [14]], "14": [39, 1, 2, 40, [15, 16]],
                     "15": [40, 1, 0, 41, [16]], "16": [41, 13, 2, 42, [17, 18]], "17": [54, 1, 0, 55, [18]],
                     "18": [55, 25, 2, 56, []]},
        "aux_on_disk": [[3, 1], [8, 7, 6, 4], [10, 9], [12, 11], [14, 13], [16, 15], [18, 17]],
        "find": {"1": 0, "9": 1, "10": 2, "11": 5, "26": 9, "27": 11, "39": 13, "40": 14, "41": 15, "42": 16,
                 "55": 17, "56": 18},
        # sample-coordinate index of every non-ref s_info entry, as listed in the same dump ("1:9:1|0" ...)
        "carrier_index": {"3": [9], "6": [10], "7": [10], "8": [10], "10": [26], "12": [28], "14": [41], "16": [42],
                          "18": [55]},
        "sampleid_map_x": "x 1001\nx 2\n1 1\nref 0\n"},
}

if __name__ == "__main__":
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "survey_vectors.json"), "w") as f:
        json.dump(vectors, f, indent=1, sort_keys=True)
