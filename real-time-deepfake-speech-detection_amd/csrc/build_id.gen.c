const char afx_build_id_str[] = "725db11312d5";
